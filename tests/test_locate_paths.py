"""The Gaussian locate has two interchangeable device paths -- per-(cell, octant) candidate lists fused with the deposit (default on a
lattice block) and the plain tree walk for every particle (FOAMYADE_NO_LOCATE_LISTS=1; also what an explicit tree, a general mesh or a
failed table allocation gets).  The switch is read when an object is created, but the library is loaded once per process with the
default, so the alternative runs the particle parity tests in a child interpreter: goldens of the reference, oracle chains bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_particle_parity_on_the_tree_walk_path():
    env = dict(os.environ, FOAMYADE_NO_LOCATE_LISTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_particle_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "set_particle_action or seeded or lattice or lists_equal"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("cap", ["2", "5", "wide"])
def test_explicit_tree_walk_on_a_short_stack(cap):
    """the explicit-tree walk (graded blocks, general meshes) keeps its DFS stack in LDS, as deep as 99.8 % of the walks sampled so far have needed (more waves per CU);
    a walk that needs more is given up and walked again by a second launch with the full depth.  FOAMYADE_LOCATE_STACK forces a depth: with 2 or 5 entries most walks
    overflow -- the golden vectors of the reference on graded meshes and the restatement's chains on a general mesh must come out bit for bit all the same.
    "wide": the 16-byte stack entries with the exact df2 that trees of 2^25 nodes and more get (FOAMYADE_LOCATE_WIDE=1) instead of the 8-byte ones with its lower bound"""
    env = dict(os.environ, FOAMYADE_LOCATE_WIDE="1") if cap == "wide" else dict(os.environ, FOAMYADE_LOCATE_STACK=cap)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_graded_mesh.py"), os.path.join(HERE, "test_ldu_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "product_on_a_graded_mesh or with_a_cloud or gaussian or point_force"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_adaptive_stack_depth_follows_the_cloud(product, oracle):
    """the depth the explicit walk's LDS stack is given comes from a histogram the kernel keeps (one walk in 64) and the host reads back a step late, renewed every 32 steps:
    a cloud of 100 000 particles on a distorted mesh (explicit k-d nodes) over 40 steps -- full depth first, then the histogram's depth, across a window's end, with a
    population that moves from one corner's fine cells to the whole box -- gives the restatement's chains bit for bit at every checked step"""
    import golden_cases as gc
    n = 24
    rs = np.random.RandomState(5)
    g = (np.arange(n) + 0.5) / n
    X, Y, Z = np.meshgrid(g ** 1.6, g ** 1.3, g, indexing="ij")                      # graded in x and y: cell sizes from 0.002 to 0.07
    C = np.stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")], axis=1)
    C += (rs.rand(*C.shape) - 0.5) * 0.2 / n                                          # and jittered: nothing lattice-like left
    V = np.full(len(C), 1.0 / n ** 3)
    lo, hi = np.zeros(3), np.ones(3)
    Nc = len(C)
    fields = dict(U=rs.rand(Nc, 3), gradP=rs.rand(Nc, 3), vGrad=rs.rand(Nc, 9), divT=rs.rand(Nc, 3), ddtU=rs.rand(Nc, 3))
    mut = dict(uSourceDrag=np.zeros(Nc), alpha=np.ones(Nc), uSource=np.zeros((Nc, 3)), uParticle=np.zeros((Nc, 3)))
    mesh = product.GeneralMesh(C, V, lo, hi)
    fy = product.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], (0, 0, -9.81),
                          mut["uSourceDrag"], mut["alpha"], mut["uSource"], mut["uParticle"], True)
    fy.setScalarProperties(2500.0, 1000.0, 1e-6)
    om = oracle.Mesh(n, n, n, 1.0 / n, (0, 0, 0), centres=C, volumes=V, bbmin=lo, bbmax=hi)
    assert np.array_equal(fy.tree_preorder(), om.pre)
    npart = 100000
    depths = []
    for step in range(40):
        rec = np.zeros((npart, 10))
        ext = 0.25 if step < 20 else 1.0                                              # the fine corner first, then everywhere
        rec[:, 0:3] = rs.rand(npart, 3) * ext
        rec[:, 9] = 1e-4
        fy.setParticles([rec])
        fy.setParticleAction(1e-3)
        if step in (0, 1, 5, 19, 20, 21, 31, 32, 33, 39):
            k, ids, w, chain = fy.stencils(0)
            sub = rs.choice(npart, 4000, replace=False)
            rk, rids, rchain, _ = oracle.range_search(C, om.pre, rec[sub, 0:3], fy.interpRange)
            assert np.array_equal(chain[sub], rchain), step
            assert np.array_equal(k[sub], np.minimum(rk, ids.shape[1])) and np.array_equal(ids[sub], rids[:, :ids.shape[1]]), step
        depths.append(fy.locate_stack_depth)
        fy.setSourceZero()
    fy.close()
    assert depths[0] == 0 and min(depths[3:]) >= 4 and max(depths) < 20, depths            # measured first, a short stack from then on
    assert len(set(depths[3:])) > 1, depths                                                 # ... and not the same one for the corner's fine cells and the whole box
