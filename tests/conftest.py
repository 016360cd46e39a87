import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_NAMES = ["example_simple", "test_scene", "metal_glass_room", "gpu_showcase", "test_comprehensive"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name: str) -> str:
    return os.path.join(SCENES, name + ".json")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ora

    ora.lib()
    return ora


@pytest.fixture(scope="session")
def gpu_ctx():
    """One pt_ctx for the whole GPU session; fails loudly if the HIP library is missing."""
    # some gpu tests hand torch device memory to the C ABI: torch must load ITS libamdhip64.so.7 before
    # libptcore.so is loaded, so both share one HIP runtime (the loader matches the SONAME)
    import torch  # noqa: F401

    from path_trace_golang_amd import build, capi

    build.build_all()  # no-op when the in-tree artefacts are current (hipcc exists on the GPU box too)
    capi.load()
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    ctx = capi.Context(ndev=1)
    yield ctx
    ctx.close()
