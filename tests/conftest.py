"""pytest config: registers the `gpu` marker and makes `my_slam_amd` (directory my-slam_amd/,
hyphenated) and the oracle binding importable."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _load_pkg():
    if "my_slam_amd" in sys.modules:
        return sys.modules["my_slam_amd"]
    spec = importlib.util.spec_from_file_location(
        "my_slam_amd", os.path.join(ROOT, "my-slam_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "my-slam_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["my_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


_load_pkg()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orbx():
    return _load_pkg()


@pytest.fixture(scope="session")
def synth():
    import my_slam_amd.synth as s
    return s
