import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_product():
    """import the product package (directory name has hyphens, so go through importlib)."""
    name = "yade_openfoam_coupling_amd"
    if name in sys.modules:
        return sys.modules[name]
    try:                                  # some tests hand torch tensors to the library: torch (its HIP runtime) first, see _share_torch_hip_runtime
        import torch  # noqa: F401
    except ImportError:
        pass
    pkg_dir = os.path.join(ROOT, "yade-openfoam-coupling_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def product():
    return load_product()


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc  # oracle/oracle.py -- the checker, never the thing under test
    orc.build()
    return orc
