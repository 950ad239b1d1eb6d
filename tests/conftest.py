import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd missing)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def api(pkg):
    """libraytrace_hip.so bound through ctypes (loading needs no GPU)."""
    if not os.path.exists(pkg.LIB_PATH):
        graft.build()
    return pkg.load_library()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle — the checker."""
    return graft.load_oracle()


def render(pkg, lib, tracer, cfg, w, h, frames, seed=1, scene_kw=None, tweak=None):
    """Drive a tracer through the RayComputeManager mirror; returns (accumulated, manager)."""
    sc = pkg.scenes.get(cfg, **(scene_kw or {}))
    mgr = sc.make_manager(tracer, lib, w, h)
    if tweak:
        tweak(mgr)
    mgr.OnEnable(renderSeed=seed)
    mgr.RenderFrames(frames)
    return tracer.read_accumulated(), mgr
