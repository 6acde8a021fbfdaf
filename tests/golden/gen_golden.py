#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own particle path in this container.

Needs /root/reference (read-only) and MPICH under /opt/conda; neither exists on the GPU box, which only
ever reads the committed .npz files.  Recipe:
    make -C oracle ref                      # compiles /root/reference/FoamYade/{FoamYade.C,meshtree/meshTree.C}
    python tests/golden/gen_golden.py       # runs oracle/_ref/ref_driver under mpiexec MPMD, packs fixtures
A fixture holds inputs (records, scalars; sha256 of the regenerable polynomial fluid fields) and the
reference's outputs (tree preorder, per-particle stencil/weights/force, the four mutable cell fields stored
sparsely, and what the fake Yade ranks received over the wire).  No reference source text is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")
MAXK = 16


def run_reference(c: gc.Case, records, workdir):
    fields = gc.fluid_fields(c)
    meta = [c.nx, c.ny, c.nz, repr(c.dx), repr(c.origin[0]), repr(c.origin[1]), repr(c.origin[2]), c.gaussian,
            c.n_yade, c.nsteps, repr(c.rhoP), repr(c.rhoF), repr(c.nu), repr(c.dt), repr(c.g[0]), repr(c.g[1]), repr(c.g[2]), c.fibre]
    with open(os.path.join(workdir, "meta.txt"), "w") as f:
        f.write(" ".join(str(m) for m in meta) + "\n")
    for name, arr in fields.items():
        np.ascontiguousarray(arr, dtype=np.float64).tofile(os.path.join(workdir, name + ".bin"))
    if gc.is_graded(c):          # non-uniform mesh: the driver takes mesh.C(), mesh.V(), mesh.points() from these instead of building a uniform block
        gc.cell_centres(c).tofile(os.path.join(workdir, "mesh_centres.bin"))
        gc.cell_volumes(c).tofile(os.path.join(workdir, "mesh_volumes.bin"))
        gc.mesh_points(c).tofile(os.path.join(workdir, "mesh_points.bin"))
    for s, R in enumerate(records):
        np.ascontiguousarray(R, dtype=np.float64).tofile(os.path.join(workdir, f"records_s{s}.bin"))
    cmd = [MPIEXEC, "-n", str(c.n_yade), DRIVER, workdir, ":", "-n", "1", DRIVER, workdir]
    subprocess.run(cmd, check=True, timeout=600, stdout=subprocess.DEVNULL)


def rd(workdir, name, dtype, shape=None):
    a = np.fromfile(os.path.join(workdir, name), dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def sparse(a, default):
    """store a mostly-constant cell field as (idx, values)"""
    a2 = a.reshape(a.shape[0], -1)
    idx = np.nonzero(np.any(a2 != default, axis=1))[0].astype(np.int32)
    return idx, a[idx]


def build_case(c: gc.Case):
    records = [gc.particle_records(c, s) for s in range(c.nsteps)]
    with tempfile.TemporaryDirectory() as wd:
        # pass 1: learn the tree root so that a probe particle can be put inside the root cell (quirk Q2:
        # the root can never enter the queue => k = 0 => "not found", meshTree.C:156-157,192)
        run_reference(c, records, wd)
        pre = rd(wd, "tree_preorder.bin", np.int32)
        root = int(pre[0])
        C = gc.cell_centres(c)
        for R in records:
            R[4, 0:3] = C[root] + np.array([0.1, -0.2, 0.15]) * c.dx       # inside the root cell
            R[5, 0:3] = C[root] + np.array([1.0, 0.1, 0.1]) * c.dx         # in its +x neighbour
        run_reference(c, records, wd)
        out = {}
        Nc = c.ncells
        out["tree_preorder"] = rd(wd, "tree_preorder.bin", np.int32)
        out["interp_scalars"] = rd(wd, "interp_scalars.bin", np.float64)
        init_alpha = rd(wd, "init_alpha.bin", np.float64)
        init_uS = rd(wd, "init_uSource.bin", np.float64)
        assert np.all(init_alpha == 1.0) and np.all(init_uS == 0.0)        # FoamYade.C:56-68
        for s in range(c.nsteps):
            Np = records[s].shape[0]
            out[f"records_s{s}"] = records[s]
            out[f"nn_s{s}"] = rd(wd, f"part_nn_s{s}.bin", np.int32)          # meshTree::nearestCell of every record (kept in nn_<case>.npz)
            out[f"k_s{s}"] = rd(wd, f"part_k_s{s}.bin", np.int32).astype(np.int8)
            out[f"incell_s{s}"] = rd(wd, f"part_incell_s{s}.bin", np.int32)
            out[f"ids_s{s}"] = rd(wd, f"part_ids_s{s}.bin", np.int32, (Np, MAXK))
            out[f"w_s{s}"] = rd(wd, f"part_w_s{s}.bin", np.float64, (Np, MAXK))
            out[f"force_s{s}"] = rd(wd, f"part_force_s{s}.bin", np.float64, (Np, 6))
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSourceDrag", 1, 0.0), ("uParticle", 3, 0.0), ("uSource", 3, 0.0)):
                a = rd(wd, f"{nm}_s{s}.bin", np.float64, (Nc, comps) if comps > 1 else (Nc,))
                idx, val = sparse(a, dflt)
                out[f"{nm}_idx_s{s}"] = idx
                out[f"{nm}_val_s{s}"] = val
            if c.gaussian:
                # optional force models (Gaussian torque + added mass) applied ON TOP of the step, see ref_driver.cpp
                out[f"forcex_s{s}"] = rd(wd, f"part_forcex_s{s}.bin", np.float64, (Np, 6))
                idx, val = sparse(rd(wd, f"uSourcex_s{s}.bin", np.float64, (Nc, 3)), 0.0)
                out[f"uSourcex_idx_s{s}"] = idx
                out[f"uSourcex_val_s{s}"] = val
            out[f"foam_yadedt_s{s}"] = rd(wd, f"foam_yadedt_s{s}.bin", np.float64)
            out[f"wire_fluiddt_s{s}"] = rd(wd, f"wire_fluiddt_s{s}.bin", np.float64)
            if c.n_yade == 1:
                out[f"wire_owner_s{s}"] = rd(wd, f"wire_owner_s{s}.bin", np.int32)
                out[f"wire_force_s{s}"] = rd(wd, f"wire_force_s{s}.bin", np.float64, (Np, 6))
            else:
                W = c.n_yade - 1
                found = np.zeros(Np, dtype=np.int32)
                wf = np.zeros((Np, 6))
                for w in range(W):
                    lo, hi = gc.split_range(Np, W, w)
                    if hi > lo:
                        found[lo:hi] = rd(wd, f"wire_found_w{w + 1}_s{s}.bin", np.int32)
                        wf[lo:hi] = rd(wd, f"wire_force_w{w + 1}_s{s}.bin", np.float64, (hi - lo, 6))
                out[f"wire_found_s{s}"] = found
                out[f"wire_force_s{s}"] = wf
        if c.n_yade > 1:
            out["wire_bbox"] = rd(wd, "wire_bbox.bin", np.float64)
        # state after setSourceZero (FoamYade.C:556-566)
        za = rd(wd, "zero_alpha.bin", np.float64)
        zs = rd(wd, "zero_uSource.bin", np.float64)
        assert np.all(zs == 0.0)
        if c.gaussian:
            assert np.all(za == 1.0)
            assert np.all(rd(wd, "zero_uSourceDrag.bin", np.float64) == 0.0)
            assert np.all(rd(wd, "zero_uParticle.bin", np.float64) == 0.0)
        out["zero_alpha_is_one"] = np.array([int(np.all(za == 1.0))], dtype=np.int32)
    fields = gc.fluid_fields(c)
    out["field_sha"] = np.array([gc.sha(fields[n]) for n in ("U", "gradP", "divT", "ddtU", "vGrad")])
    out["centres_sha"] = np.array([gc.sha(gc.cell_centres(c))])
    if gc.is_graded(c):
        out["volumes_sha"] = np.array([gc.sha(gc.cell_volumes(c))])
    return out


def main():
    if not os.path.exists(DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    nn_only = "--nn-only" in sys.argv          # only the nearestCell fixtures (added in round 2; the other files stay as committed)
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or [c.name for c in gc.CASES + gc.FIBRE_CASES + gc.GRADED_CASES]
    for name in names:
        c = gc.CASES_BY_NAME[name]
        out = build_case(c)
        nn = {k: out.pop(k) for k in [k for k in out if k.startswith("nn_s")]}
        nn["records_sha"] = np.array([gc.sha(out[f"records_s{s}"]) for s in range(c.nsteps)])
        np.savez_compressed(os.path.join(HERE, "nn_" + name + ".npz"), **nn)
        if nn_only:
            print(f"{name}: nearestCell of {sum(v.size for k, v in nn.items() if k.startswith('nn_s'))} points")
            continue
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        k = out["k_s0"].astype(int)
        print(f"{name}: Np={k.size} found={int((k > 0).sum())} mean_k={k[k > 0].mean() if (k > 0).any() else 0:.3f} "
              f"max_k={k.max()} size={os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
