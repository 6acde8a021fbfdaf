#!/usr/bin/env python3
"""The three turbulence closures at the C3 size (160^3 cells, 10 M particles): ms per coupled step next to the laminar run (development check)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402

prod = ge.load_product()
n, npart = 160, 10_000_000
rec = bench.c3_particles(torch, npart, n, 3, torch.device("cuda", 0))
for name, kw in (("laminar", {}), ("Smagorinsky", dict(turbulence_model=1, nut_initial=1e-6)),
                 ("kEqn", dict(turbulence_model=2, nut_initial=1e-6, k_initial=1e-5)),
                 ("kEpsilon", dict(turbulence_model=3, nut_initial=1e-6, k_initial=1e-5, eps_initial=1e-4))):
    case = bench.c3_case(prod, n, 1e-4, 1, 1)
    for k, v in kw.items():
        setattr(case, k, v)
    s = prod.Solver(case)
    s.set_particles_device(rec)
    s.enable_particle_timing(True)
    for _ in range(2):
        s.step()
    acc = 0.0
    for _ in range(5):
        s.step()
        acc += s.stats()["ms_total"] / 5
    extra = ""
    if kw:
        nut = s.get("nut")
        extra = f" nut in [{nut.min():.3g}, {nut.max():.3g}]"
    print(f"{name:12s} {acc:7.3f} ms per step{extra}")
    s.close()
