"""Host restatement of the candidate-list locate (tools/proto/locate_lists.cpp: the builder of k_build_locate_lists and the scan of
k_locate_lists / k_locate_deposit, same margins) against the plain k-d walk: for sampled (cell, octant) lists -- boundary cells
included -- random queries, queries at the hand-over distance from the cell faces, next to and exactly on the octant planes must give
identical improvement chains (ids and squared distances, bit for bit).  The GPU tests then pin the device kernels against the
reference's golden vectors and the oracle (tests/test_particle_parity.py)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def proto(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("proto") / "locate_lists")
    csrc = os.path.join(ROOT, "yade-openfoam-coupling_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + csrc, os.path.join(ROOT, "tools", "proto", "locate_lists.cpp"), os.path.join(csrc, "kdtree.cpp"),
                    "-o", exe, "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("shape,cells,env", [((40, 40, 40), 400, {}), ((24, 18, 10), 400, {}), ((9, 7, 5), 150, {}), ((48, 40, 33), 300, {"BIGO": "1"})])
def test_list_scan_equals_the_walk(proto, shape, cells, env):
    r = subprocess.run([proto, *map(str, shape), str(cells)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"mean len ([\d.]+), max (\d+).*queries (\d+), mismatches (\d+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(4)) == 0 and int(m.group(3)) >= 64 * 8 * cells
    assert float(m.group(1)) < 9.0 and int(m.group(2)) <= 24          # what kLocateListLen = 24 is sized for
