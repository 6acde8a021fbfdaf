"""The HEADLINE configurations against the oracle at their own size (task 2): two coupled steps of C3 (160^3 cells,
10 M particles, pimpleFoamYade Gaussian 4-way) and of C2 (200 x 100 x 50 cells, 1 M particles, icoFoamYade point force) on the HIP
path (through the C-ABI) and on the CPU oracle, from the case's initial state, on the records bench.py times (same generator calls).

Two steps, because the first step of the BASELINE cloud is degenerate (U = p = 0 and particles at rest: every force is zero); the second
sees the hydrostatic pressure the first one built, i.e. a non-trivial Archimedes force and momentum source.  `c3_moving` gives the same
cloud velocities (+-0.05 m/s) so that the drag terms and uSourceDrag are exercised at that size as well.

Bars: index work (stencil sizes k, stencil cell ids, improvement-chain lengths, found flags) bit-exact; after the FIRST step (exact inputs on both
sides) alpha / uSource / uSourceDrag and the per-particle forces 1e-10 of the array's largest magnitude (same operations, different summation
order in the deposits); after the second step alpha still 1e-10, forces / sources and U, p, phi to the FV tolerance (the linear solvers stop on a residual; reduction order differs: 1e-5 as in test_fv_parity.py); PCG iterations +- 2.
FV parity is UNPINNED against the reference itself (OpenFOAM-6 is not in the image): this test shows HIP == oracle, not oracle == OpenFOAM."""
import os

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

RTOL_PARTICLE = 1e-10
RTOL_FV = 1e-5


@pytest.mark.parametrize("config,velocities", [("c3", False), ("c3", True), ("c2", False)])
def test_headline_configuration_matches_oracle_at_full_size(product, oracle, config, velocities):
    import torch
    n, n_part, dt = (100, 1_000_000, 2e-3) if config == "c2" else (160, 10_000_000, 1e-4)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    rec = bench.bench_records_host(torch, config, n, n_part, velocities=velocities)
    _, cells, ref = bench.oracle_steps(oracle, config, n, dt, threads, rec, collect=True, steps=2)
    assert cells == (1_000_000 if config == "c2" else 4_096_000)
    case = bench.c2_case(product, dt, 1) if config == "c2" else bench.c3_case(product, n, dt, 1)
    res = bench.hip_vs_oracle(product, case, rec, ref, device=0, steps=2)
    print(res)
    assert res["found_equal"]
    # step 1: exact inputs on both sides (the initial fields) -- the particle side at the particle tolerance
    s1 = res["step1"]
    assert s1["force"] <= RTOL_PARTICLE, s1
    for nm in (("alpha", "uSource", "uSourceDrag") if config != "c2" else ("uSource",)):
        assert s1[nm] <= RTOL_PARTICLE, (nm, s1)
    if velocities:
        assert s1["force_scale"] > 0.0                            # (a moving cloud in a fluid at rest: drag from the first step on)
    if config != "c2":
        assert res["k_equal"] and res["ids_equal"] and res["chain_equal"]
        assert res["pairs"] > 5 * n_part
        assert res["alpha"] <= RTOL_PARTICLE, res["alpha"]        # (the same cloud: the void fraction does not depend on the fluid's state)
        assert res["force_scale"] > 0.0                           # (the second step's forces are not identically zero: Archimedes in the hydrostatic field)
    # step 2: forces and sources gather U / grad p / the stress divergence, which carry the FV tolerance after a step
    for nm in (("uSource", "uSourceDrag") if config != "c2" else ("uSource",)):
        assert res[nm] <= RTOL_FV, (nm, res[nm])
    assert res["force"] <= RTOL_FV, res["force"]
    for nm in ("U", "p", "phi_x", "phi_y", "phi_z"):
        assert res[nm] <= RTOL_FV, (nm, res[nm])
    assert abs(res["p_iters"][0] - res["p_iters"][1]) <= 2, res["p_iters"]
    assert abs(res["u_iters"][0] - res["u_iters"][1]) <= 1, res["u_iters"]
