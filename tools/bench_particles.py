#!/usr/bin/env python3
"""Particle-phase-only timing at BASELINE configs (development tool; bench.py is the judged benchmark)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=160)
    ap.add_argument("--np", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--gaussian", type=int, default=1)
    ap.add_argument("--fill", type=float, default=0.6)
    ap.add_argument("--vel", type=float, default=0.0, help="particle velocity scale (0: at rest, the C3 cloud)")
    ap.add_argument("--mesh", default="block", help="block | wavy: the block's centres displaced by 0.6 dx sin sin sin, handed over as a general mesh (explicit k-d tree: "
                    "k_locate<false>, k_deposit)")
    a = ap.parse_args()
    import torch
    prod = ge.load_product()
    n = a.n
    dx = 1.0 / n
    t0 = time.time()
    mesh = prod.BlockMesh(n, n, n, dx)
    if a.mesh == "wavy":
        P = mesh.C
        sw = np.sin(np.pi * P[:, 0]) * np.sin(np.pi * P[:, 1]) * np.sin(np.pi * P[:, 2])
        amp = 0.6 * dx
        Q = P.copy()
        Q[:, 0] += amp * sw * np.cos(30 * P[:, 1]); Q[:, 1] += amp * sw * np.cos(20 * P[:, 2] + 1); Q[:, 2] += amp * sw * np.cos(40 * P[:, 0] + 2)
        mesh = prod.GeneralMesh(Q, mesh.V, mesh.bbox_min, mesh.bbox_max)
    Nc = mesh.n_cells
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    U = (torch.rand(Nc, 3, dtype=torch.float64, generator=g) * 0.1).to(dev)
    gradP = (torch.rand(Nc, 3, dtype=torch.float64, generator=g) * 100.0).to(dev)
    divT = (torch.rand(Nc, 3, dtype=torch.float64, generator=g)).to(dev)
    vGrad = (torch.rand(Nc, 9, dtype=torch.float64, generator=g)).to(dev)
    ddtU = torch.zeros(Nc, 3, dtype=torch.float64, device=dev)
    uSourceDrag = torch.zeros(Nc, dtype=torch.float64, device=dev)
    alpha = torch.zeros(Nc, dtype=torch.float64, device=dev)
    uSource = torch.zeros(Nc, 3, dtype=torch.float64, device=dev)
    uParticle = torch.zeros(Nc, 3, dtype=torch.float64, device=dev)
    rec = torch.rand(a.np, 10, dtype=torch.float64, generator=g)
    rec[:, 2] *= a.fill
    rec[:, 3:9] = 0.0
    if a.vel:
        rec[:, 3:6] = (torch.rand(a.np, 3, dtype=torch.float64, generator=g) - 0.5) * 2 * a.vel
    rec[:, 9] = 0.2 * dx
    rec = rec.to(dev).contiguous()
    print(f"inputs ready {time.time() - t0:.1f}s", flush=True)
    t0 = time.time()
    fy = prod.FoamYade(mesh, U, gradP, vGrad, divT, ddtU, (0, 0, -9.81), uSourceDrag, alpha, uSource, uParticle, bool(a.gaussian))
    print(f"fy_create (tree build + upload) {time.time() - t0:.1f}s", flush=True)
    fy.setScalarProperties(2650.0, 1000.0, 1e-6)
    fy.enable_timing(True)
    fy.setParticlesDevice([rec])
    for s in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.time()
        fy.setParticleAction(1e-4)
        torch.cuda.synchronize()
        wall = (time.time() - t0) * 1e3
        t = fy.timings()
        print(f"step {s}: wall {wall:.2f} ms | bin {t['bin']:.2f} locate+deposit {t['locate_deposit']:.2f} finalize {t['finalize']:.2f} "
              f"force {t['force']:.2f} total {t['total']:.2f}", flush=True)
        fy.setSourceZero()
    k = fy.stencils(0)[0] if a.np <= 2_000_000 else None
    if k is not None:
        print("mean k", k[k > 0].mean(), "found", (k > 0).mean())


if __name__ == "__main__":
    main()
