"""The Gaussian locate has three interchangeable device paths -- candidate lists fused with the deposit (default), candidate lists
followed by k_deposit (FOAMYADE_UNFUSED_DEPOSIT), and the plain tree walk for every particle (FOAMYADE_NO_LOCATE_LISTS; also what an
explicit tree or a failed table allocation gets).  The switches are read once per process, so each alternative runs the particle
parity tests in a child interpreter: goldens of the reference, oracle chains bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("switch", ["FOAMYADE_UNFUSED_DEPOSIT", "FOAMYADE_NO_LOCATE_LISTS"])
def test_particle_parity_on_the_alternative_locate_paths(switch):
    env = dict(os.environ, **{switch: "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_particle_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "set_particle_action or seeded or lattice or lists_equal"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
