import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import the hyphenated package directory genome-on-diet_amd/ as module genome_on_diet_amd"""
    name = "genome_on_diet_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "genome-on-diet_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import gdo
    gdo.build_oracle()
    return gdo, gdo.load_oracle()


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    import torch  # first: share torch's HIP runtime
    assert torch.cuda.is_available(), "GPU tests need a device"
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()
