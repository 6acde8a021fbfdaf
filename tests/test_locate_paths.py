"""The Gaussian locate has two interchangeable device paths -- per-(cell, octant) candidate lists fused with the deposit (default on a
lattice block) and the plain tree walk for every particle (FOAMYADE_NO_LOCATE_LISTS=1; also what an explicit tree, a general mesh or a
failed table allocation gets).  The switch is read when an object is created, but the library is loaded once per process with the
default, so the alternative runs the particle parity tests in a child interpreter: goldens of the reference, oracle chains bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_particle_parity_on_the_tree_walk_path():
    env = dict(os.environ, FOAMYADE_NO_LOCATE_LISTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_particle_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "set_particle_action or seeded or lattice or lists_equal"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("cap", ["2", "5"])
def test_explicit_tree_walk_on_a_short_stack(cap):
    """the explicit-tree walk (graded blocks, general meshes) keeps its DFS stack in LDS, as deep as 99.8 % of the walks sampled so far have needed (more waves per CU);
    a walk that needs more is given up and walked again by a second launch with the full depth.  FOAMYADE_LOCATE_STACK forces a depth: with 2 or 5 entries most walks
    overflow -- the golden vectors of the reference on graded meshes and the restatement's chains on a general mesh must come out bit for bit all the same"""
    env = dict(os.environ, FOAMYADE_LOCATE_STACK=cap)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_graded_mesh.py"), os.path.join(HERE, "test_ldu_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "product_on_a_graded_mesh or with_a_cloud or gaussian or point_force"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
