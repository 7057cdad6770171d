"""pytest plumbing: markers, package / oracle loaders, synthetic packet helpers."""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the hyphen-named package directory esp32-opus-player_amd as esp32_opus_player_amd."""
    name = "esp32_opus_player_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "esp32-opus-player_amd", "__init__.py")
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
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()
