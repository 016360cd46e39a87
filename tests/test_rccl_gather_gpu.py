"""PTCORE_GATHER=rccl: a context collects the tiles of a frame on devices[0] through RCCL (librccl.so dlopen'ed by libptcore,
one communicator per device from ncclCommInitAll, grouped ncclSend / ncclRecv on the devices' streams) instead of
hipMemcpyPeerAsync -- the in-library form of north_star's "RCCL gather over xGMI of per-tile framebuffers" (DESIGN 7).
One GPU is all this box has, so what runs here is the plumbing: a one-rank communicator whose rank 0 sends its tiles to itself.
The frame must be the oracle's either way; tests/test_multigpu_gpu.py repeats it on two physical devices when there are two."""
import numpy as np
import pytest

from conftest import render_vs_oracle, scene_path

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_gather_gives_the_oracle_frame(monkeypatch, oracle):
    import torch  # noqa: F401  (a process that has PyTorch's RCCL gets that one: libptcore asks for the loaded library first)

    from path_trace_golang_amd import capi, scene

    monkeypatch.setenv("PTCORE_GATHER", "rccl")
    L = capi.load()
    sc = scene.load(scene_path("gpu_showcase"))
    w, h, spp, depth, seed = 100, 70, 3, 6, 5  # 4 x 3 tiles, ragged on both edges
    o = oracle.render(oracle.Scene.load(scene_path("gpu_showcase")), w, h, spp, depth, seed=seed)
    with capi.Context(ndev=1) as ctx:
        assert L.pt_debug_gather_mode(ctx.handle) == 1
        render_vs_oracle(ctx, sc, o, w, h, spp, depth, seed)  # rgba, sums and (with the flag) both counter planes through the group
        render_vs_oracle(ctx, sc, o, w, h, spp, depth, seed, forms=("shipping",))  # a communicator serves frame after frame
    monkeypatch.delenv("PTCORE_GATHER")
    with capi.Context(ndev=1) as ctx:
        assert L.pt_debug_gather_mode(ctx.handle) == 0


def test_rccl_refuses_one_gpu_listed_twice_and_bad_values_are_errors(monkeypatch):
    from path_trace_golang_amd import capi

    monkeypatch.setenv("PTCORE_GATHER", "rccl")
    with pytest.raises(capi.PtError) as e:  # the virtual devices of the other tests: RCCL wants distinct GPUs, and says so
        capi.Context(devices=[0, 0])
    assert "ncclCommInitAll" in str(e.value)
    monkeypatch.setenv("PTCORE_GATHER", "carrier-pigeon")
    with pytest.raises(capi.PtError) as e:
        capi.Context(ndev=1)
    assert "PTCORE_GATHER" in str(e.value)
    monkeypatch.delenv("PTCORE_GATHER")
    with capi.Context(ndev=1) as ctx:  # and the process is fine afterwards
        assert ctx.handle
