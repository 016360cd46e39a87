"""HIP path vs the CPU oracle on identical (seed, pixel, sample) streams.

Tolerance (stated, see DESIGN.md "Parity bar"): the paths are identical decision for
decision, so segment and RNG-draw counts per pixel must be EQUAL and the 8-bit image must be
EQUAL; the FP64 radiance sums may differ only by the association order of the throughput
product (the oracle nests a1*(a2*(...*L)) like the reference's recursion, the kernel
carries T=((a1*a2)*...)), bounded by depth * 2^-52 relative per sample.
"""
import numpy as np
import pytest

from conftest import SCENE_NAMES, render_vs_oracle, scene_path

pytestmark = pytest.mark.gpu


def _render_gpu(ctx, name, w, h, spp, depth, seed=1, chunk=0, stats=True):
    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path(name))
    cfg = hip.RenderConfig(w, h, spp, depth, seed, chunk, capi.PT_FLAG_PIXEL_STATS if stats else 0)
    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3), np.float64)
    nseg = np.zeros((h, w), np.uint32) if stats else None
    ndraw = np.zeros((h, w), np.uint32) if stats else None
    st = hip.render(sc, cfg, img, None, acc, nseg, ndraw, ctx=ctx)
    return img, acc, nseg, ndraw, st


def _compare(ora_out, img, acc, nseg, ndraw, st, depth):
    assert st["segments"] == ora_out["stats"]["segments"]
    assert st["exit_scans"] == ora_out["stats"]["exit_scans"]
    assert st["draws"] == ora_out["stats"]["draws"]
    assert st["samples"] == ora_out["stats"]["samples"]
    if nseg is not None:
        assert np.array_equal(nseg, ora_out["nseg"])
        assert np.array_equal(ndraw, ora_out["ndraw"])
    ref = ora_out["accum"]
    tol = 4.0 * max(depth, 1) * 2.0 ** -52
    err = np.abs(acc - ref)
    bound = tol * np.maximum(np.abs(ref), 1e-300)
    assert np.all(err <= bound), "max rel err %g" % float(np.max(err / np.maximum(np.abs(ref), 1e-300)))
    assert np.array_equal(img, ora_out["rgba"])


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_scene_small_matches_oracle(gpu_ctx, oracle, name):
    w, h, spp, depth = 96, 54, 8, 8
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=1)
    for stats in (True, False):  # the counting build and the one that ships
        img, acc, nseg, ndraw, st = _render_gpu(gpu_ctx, name, w, h, spp, depth, seed=1, stats=stats)
        _compare(o, img, acc, nseg, ndraw, st, depth)


def test_c1_plumbing_config(gpu_ctx, oracle):
    # BASELINE config 1: example_simple 256x256, 16 spp, depth 4
    from path_trace_golang_amd import scene

    o = oracle.render(oracle.Scene.load(scene_path("example_simple")), 256, 256, 16, 4, seed=1)
    render_vs_oracle(gpu_ctx, scene.load(scene_path("example_simple")), o, 256, 256, 16, 4, 1)


def test_chunking_does_not_change_pixels(gpu_ctx):
    a = _render_gpu(gpu_ctx, "gpu_showcase", 80, 45, 12, 6, seed=3, chunk=0, stats=False)
    b = _render_gpu(gpu_ctx, "gpu_showcase", 80, 45, 12, 6, seed=3, chunk=5, stats=False)
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1], b[1])  # sample order is preserved across chunks: bit-equal sums


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_golden_fixture(gpu_ctx, name):
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    w, h, spp, depth, seed = (int(x) for x in g["cfg"])
    for stats in (True, False):  # the counting build and the one that ships
        img, acc, nseg, ndraw, st = _render_gpu(gpu_ctx, name, w, h, spp, depth, seed=seed, stats=stats)
        assert [st["samples"], st["segments"], st["exit_scans"], st["draws"]] == [int(x) for x in g["totals"]]
        assert np.array_equal(img, g["rgba"])
        if stats:
            assert np.array_equal(nseg, g["nseg"]) and np.array_equal(ndraw, g["ndraw"])
        rel = np.abs(acc - g["accum"]) / np.maximum(np.abs(g["accum"]), 1e-300)
        assert rel.max() <= 4 * depth * 2.0 ** -52


@pytest.mark.parametrize("w,h", [(33, 31), (1, 1), (7, 100), (65, 9), (400, 225)])
def test_ragged_frame_sizes(gpu_ctx, oracle, w, h):
    spp, depth = (2, 5) if w * h > 10000 else (5, 6)
    o = oracle.render(oracle.Scene.load(scene_path("test_scene")), w, h, spp, depth, seed=2)
    for stats in (True, False):
        img, acc, nseg, ndraw, st = _render_gpu(gpu_ctx, "test_scene", w, h, spp, depth, seed=2, stats=stats)
        _compare(o, img, acc, nseg, ndraw, st, depth)


def test_deep_paths_reference_final_depth(gpu_ctx, oracle):
    # the reference's "final" preset uses depth 80 (util.go:28-33): glass-box creep paths burn all of it
    w, h, spp, depth = 40, 24, 3, 80
    o = oracle.render(oracle.Scene.load(scene_path("metal_glass_room")), w, h, spp, depth, seed=4)
    img, acc, nseg, ndraw, st = _render_gpu(gpu_ctx, "metal_glass_room", w, h, spp, depth, seed=4)
    _compare(o, img, acc, nseg, ndraw, st, depth)
    assert int(nseg.max()) > 3 * 40  # some pixel really went deep
    img2, acc2, _, _, st2 = _render_gpu(gpu_ctx, "metal_glass_room", w, h, spp, depth, seed=4, stats=False)
    _compare(o, img2, acc2, None, None, st2, depth)  # the build that ships
    assert np.array_equal(acc, acc2)


def test_depth_zero_and_one(gpu_ctx, oracle):
    for depth in (0, 1, 2, 3):
        o = oracle.render(oracle.Scene.load(scene_path("example_simple")), 40, 30, 4, depth, seed=6)
        for stats in (True, False):
            img, acc, nseg, ndraw, st = _render_gpu(gpu_ctx, "example_simple", 40, 30, 4, depth, seed=6, stats=stats)
            if depth == 0:
                # rayColorOpt returns black before any scan (renderer.go:287-289); the camera draws are still made
                assert not acc.any() and not o["accum"].any() and st["segments"] == 0 and st["draws"] > 0
            _compare(o, img, acc, nseg, ndraw, st, depth)


def test_seed_changes_image_and_is_reproducible(gpu_ctx):
    a = _render_gpu(gpu_ctx, "example_simple", 64, 36, 4, 6, seed=1, stats=False)
    b = _render_gpu(gpu_ctx, "example_simple", 64, 36, 4, 6, seed=1, stats=False)
    c = _render_gpu(gpu_ctx, "example_simple", 64, 36, 4, 6, seed=2, stats=False)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert not np.array_equal(a[1], c[1])


def test_row_stride_and_alpha(gpu_ctx, oracle):
    from path_trace_golang_amd import hip, scene

    w, h = 50, 20
    big = np.full((h, w + 7, 4), 7, np.uint8)
    view = big[:, :w]  # row stride 4*(w+7)
    hip.render(scene.load(scene_path("example_simple")), hip.RenderConfig(w, h, 2, 4, 3), view, ctx=gpu_ctx)
    o = oracle.render(oracle.Scene.load(scene_path("example_simple")), w, h, 2, 4, seed=3, want=("rgba",))
    assert np.array_equal(view, o["rgba"]) and np.all(view[..., 3] == 255)
    assert np.all(big[:, w:] == 7)  # bytes beyond each row are untouched


def test_size_mismatch_is_a_silent_no_op(gpu_ctx):
    from path_trace_golang_amd import hip, scene

    img = np.full((10, 10, 4), 9, np.uint8)
    out = hip.render(scene.load(scene_path("example_simple")), hip.RenderConfig(12, 10, 1, 2, 1), img, ctx=gpu_ctx)
    assert out == {} and np.all(img == 9)  # renderer.go:46-49


def test_invalid_arguments_are_errors(gpu_ctx):
    import ctypes as C

    from path_trace_golang_amd import capi, hip, scene

    L = capi.load()
    flat = hip.FlatScene(scene.load(scene_path("example_simple")))
    cfg = hip.pt_config(hip.RenderConfig(0, 10, 1, 1))
    st = capi.PtStats()
    buf = np.zeros((10, 10, 4), np.uint8)
    rc = L.pt_render(gpu_ctx.handle, C.byref(flat.c), C.byref(cfg), buf.ctypes.data_as(C.c_void_p), 40, None, None, None,
                     C.byref(st))
    assert rc == capi.PT_ERR_INVALID and b"width" in L.pt_last_error()
    cfg = hip.pt_config(hip.RenderConfig(10, 10, 1, 1))
    rc = L.pt_render(gpu_ctx.handle, C.byref(flat.c), C.byref(cfg), buf.ctypes.data_as(C.c_void_p), 39, None, None, None,
                     C.byref(st))
    assert rc == capi.PT_ERR_INVALID
    # 1 x 2^27 pixels would be 2^22 nearly empty tiles: refused before any pointer is touched
    cfg = hip.pt_config(hip.RenderConfig(1, 1 << 27, 1, 1))
    rc = L.pt_render(gpu_ctx.handle, C.byref(flat.c), C.byref(cfg), buf.ctypes.data_as(C.c_void_p), 4, None, None, None,
                     C.byref(st))
    assert rc == capi.PT_ERR_INVALID and b"too large" in L.pt_last_error()
    assert L.pt_step(gpu_ctx.handle, 1, None) == capi.PT_ERR_STATE  # no frame open
    # a failed call leaves the context usable
    img = np.zeros((10, 10, 4), np.uint8)
    assert hip.render(scene.load(scene_path("example_simple")), hip.RenderConfig(10, 10, 1, 2, 1), img, ctx=gpu_ctx)["samples"] == 100


def test_progressive_steps_equal_one_shot(gpu_ctx):
    import ctypes as C

    from path_trace_golang_amd import capi, hip, scene

    L = capi.load()
    sc = scene.load(scene_path("gpu_showcase"))
    w, h, spp, depth = 64, 40, 10, 6
    one_img = np.zeros((h, w, 4), np.uint8)
    one_acc = np.zeros((h, w, 3))
    hip.render(sc, hip.RenderConfig(w, h, spp, depth, 8), one_img, None, one_acc, ctx=gpu_ctx)
    calls = []
    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3))
    st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 8), img, lambda: calls.append(img.copy()), acc, ctx=gpu_ctx)
    assert len(calls) == 10 + 1  # every spp/10 samples + the final refresh (gpu.go:2209-2212, :2523-2525)
    assert np.array_equal(img, one_img) and np.array_equal(acc, one_acc)
    assert st["samples"] == w * h * spp
    # an intermediate read is the estimate normalised by the samples done so far
    flat = hip.FlatScene(sc)
    cfg = hip.pt_config(hip.RenderConfig(w, h, spp, depth, 8))
    capi.check(L.pt_begin(gpu_ctx.handle, C.byref(flat.c), C.byref(cfg)))
    done = C.c_int32()
    capi.check(L.pt_step(gpu_ctx.handle, 4, C.byref(done)))
    assert done.value == 4
    part = np.zeros((h, w, 4), np.uint8)
    pacc = np.zeros((h, w, 3))
    capi.check(L.pt_read(gpu_ctx.handle, part.ctypes.data_as(C.c_void_p), w * 4, pacc.ctypes.data_as(C.c_void_p)))
    capi.check(L.pt_end(gpu_ctx.handle, None))
    four = np.zeros((h, w, 4), np.uint8)
    facc = np.zeros((h, w, 3))
    hip.render(sc, hip.RenderConfig(w, h, 4, depth, 8), four, None, facc, ctx=gpu_ctx)
    # samples 0..3 of the 10-spp render are the 4-spp render (streams are keyed by sample index)...
    assert np.array_equal(pacc, facc)
    # ...but depth-3 roulette etc. is per path, so the images agree too
    assert np.array_equal(part, four)


def test_two_virtual_devices_in_process(oracle):
    # the in-process multi-device path (interleaved tiles, peer-copy gather on devices[0], untile) with the
    # same physical GPU listed twice and three times: pixels must not depend on the device count
    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path("test_comprehensive"))
    w, h, spp, depth = 100, 70, 3, 6
    o = oracle.render(oracle.Scene.load(scene_path("test_comprehensive")), w, h, spp, depth, seed=5)
    for devs in ([0, 0], [0, 0, 0]):
        with capi.Context(devices=devs) as ctx:
            img = np.zeros((h, w, 4), np.uint8)
            acc = np.zeros((h, w, 3))
            nseg = np.zeros((h, w), np.uint32)
            ndraw = np.zeros((h, w), np.uint32)
            st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 5, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg,
                            ndraw, ctx=ctx)
            assert st["num_devices"] == len(devs)
            _compare(o, img, acc, nseg, ndraw, st, depth)
            img2 = np.zeros((h, w, 4), np.uint8)
            acc2 = np.zeros((h, w, 3))
            st2 = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 5), img2, None, acc2, ctx=ctx)  # the build that ships
            _compare(o, img2, acc2, None, None, st2, depth)
    # more devices than tiles: a 40x20 frame is two tiles, the third and fourth device own nothing
    w, h = 40, 20
    o = oracle.render(oracle.Scene.load(scene_path("test_comprehensive")), w, h, spp, depth, seed=5)
    with capi.Context(devices=[0, 0, 0, 0]) as ctx:
        img = np.zeros((h, w, 4), np.uint8)
        acc = np.zeros((h, w, 3))
        nseg = np.zeros((h, w), np.uint32)
        ndraw = np.zeros((h, w), np.uint32)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 5, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw,
                        ctx=ctx)
        _compare(o, img, acc, nseg, ndraw, st, depth)


def test_device_tiles_and_untile_entry_points(gpu_ctx, oracle):
    # the one-process-per-GPU path used by bench.py, with the shards rendered one after another
    import ctypes as C

    import torch

    from path_trace_golang_amd import capi, hip, scene, tiling

    L = capi.load()
    w, h, spp, depth, world = 100, 70, 2, 5, 3
    flat = hip.FlatScene(scene.load(scene_path("gpu_showcase")))
    cfg = hip.pt_config(hip.RenderConfig(w, h, spp, depth, 9))
    stride = tiling.max_shard_tiles(w, h, world)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    bufs, abufs = [], []
    for k in range(world):
        t = torch.zeros(stride * 4096, dtype=torch.uint8, device=dev)
        a = torch.zeros(stride * 3072, dtype=torch.float64, device=dev)
        sh = capi.PtShard(k, world)
        st = capi.PtStats()
        capi.check(L.pt_render_tiles_device(gpu_ctx.handle, C.byref(flat.c), C.byref(cfg), C.byref(sh),
                                            C.c_void_p(t.data_ptr()), C.c_void_p(a.data_ptr()),
                                            C.c_void_p(stream.cuda_stream), C.byref(st)))
        assert st.samples == sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in
                                 (tiling.tile_rect(w, h, tt) for tt in tiling.shard_tiles(w, h, k, world))) * spp
        bufs.append(t)
        abufs.append(a)
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    facc = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
    packed, apacked = torch.cat(bufs), torch.cat(abufs)  # keep alive until the kernel has run
    capi.check(L.pt_untile_device(gpu_ctx.handle, w, h, world, stride, C.c_void_p(packed.data_ptr()),
                                  C.c_void_p(apacked.data_ptr()), C.c_void_p(frame.data_ptr()), w * 4,
                                  C.c_void_p(facc.data_ptr()), C.c_void_p(stream.cuda_stream)))
    torch.cuda.synchronize()
    o = oracle.render(oracle.Scene.load(scene_path("gpu_showcase")), w, h, spp, depth, seed=9, want=("rgba", "accum"))
    assert np.array_equal(frame.cpu().numpy(), o["rgba"])
    # and the device buffers agree with the numpy tiling convention
    assert np.array_equal(tiling.untile([b.cpu().numpy() for b in bufs], w, h, world, 4, np.uint8, stride), o["rgba"])
    rel = np.abs(facc.cpu().numpy() - o["accum"]) / np.maximum(np.abs(o["accum"]), 1e-300)
    assert rel.max() <= 4 * depth * 2.0 ** -52


def test_full_size_properties_c2(gpu_ctx):
    # BASELINE config 2 at full size (800x600, 256 spp, depth 8): size-independent checks, no oracle
    from path_trace_golang_amd import hip, scene

    sc = scene.load(scene_path("test_scene"))
    w, h, spp, depth = 800, 600, 256, 8
    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3))
    st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 1), img, None, acc, ctx=gpu_ctx)
    assert st["samples"] == w * h * spp and w * h * spp <= st["segments"] <= w * h * spp * depth
    assert np.all(img[..., 3] == 255) and np.all(np.isfinite(acc)) and np.all(acc >= 0)
    # the 8-bit image is exactly the pixel finish of the sums (renderer.go:190-221)
    v = np.sqrt(acc * (1.0 / spp)) * 255.999
    q = np.clip(v, 0, 255.999).astype(np.uint8)
    assert np.array_equal(q, img[..., :3])
    # chunked == unchunked at full size (ordered accumulation)
    img2 = np.zeros((h, w, 4), np.uint8)
    acc2 = np.zeros((h, w, 3))
    hip.render(sc, hip.RenderConfig(w, h, spp, depth, 1, 37), img2, None, acc2, ctx=gpu_ctx)
    assert np.array_equal(acc, acc2) and np.array_equal(img, img2)
    # halves of the sample range add up: spp 0..127 rendered alone is a prefix of the stream set
    acc_half = np.zeros((h, w, 3))
    hip.render(sc, hip.RenderConfig(w, h, 128, depth, 1), np.zeros((h, w, 4), np.uint8), None, acc_half, ctx=gpu_ctx)
    assert np.all(acc_half <= acc + 1e-9)


FULL_SIZE = [  # BASELINE configs 3, 4, 5 at FULL size, with windows of the frame for the oracle
    ("metal_glass_room", 1920, 1080, 1024, 12, [(930, 560, 962, 576), (1900, 1064, 1920, 1080)]),
    ("gpu_showcase", 1920, 1080, 1024, 8, [(944, 520, 976, 536), (1900, 1064, 1920, 1080), (0, 0, 16, 8)]),
    ("test_comprehensive", 3840, 2160, 4096, 16, [(1900, 1100, 1916, 1108), (3824, 2152, 3840, 2160)]),
]


@pytest.mark.parametrize("name,w,h,spp,depth,windows", FULL_SIZE, ids=["C3", "C4", "C5"])
def test_full_size_windows_match_oracle(gpu_ctx, oracle, name, w, h, spp, depth, windows):
    # The oracle renders windows of the very frame the configuration names (same streams: they are keyed by
    # pixel and sample), which must match the GPU frame exactly; plus size-independent properties of the frame.
    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path(name))
    seed = 1
    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3))
    nseg = np.zeros((h, w), np.uint32)
    ndraw = np.zeros((h, w), np.uint32)
    st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw,
                    ctx=gpu_ctx)
    assert st["samples"] == w * h * spp
    assert int(nseg.sum(dtype=np.uint64)) == st["segments"] and int(ndraw.sum(dtype=np.uint64)) == st["draws"]
    assert np.all(img[..., 3] == 255) and np.all(np.isfinite(acc)) and np.all(acc >= 0)
    q = np.clip(np.sqrt(acc * (1.0 / spp)) * 255.999, 0, 255.999).astype(np.uint8)
    assert np.array_equal(q, img[..., :3])
    osc = oracle.Scene.load(scene_path(name))
    for x0, y0, x1, y1 in windows:
        o = oracle.render(osc, w, h, spp, depth, seed=seed, window=(x0, y0, x1, y1))
        sl = (slice(y0, y1), slice(x0, x1))
        assert np.array_equal(nseg[sl], o["nseg"][sl]) and np.array_equal(ndraw[sl], o["ndraw"][sl])
        assert np.array_equal(img[sl], o["rgba"][sl])
        rel = np.abs(acc[sl] - o["accum"][sl]) / np.maximum(np.abs(o["accum"][sl]), 1e-300)
        assert rel.max() <= 4 * depth * 2.0 ** -52
    # The same frame from the kernels that SHIP (no PT_FLAG_PIXEL_STATS: other instantiations -- the six-wave split kernel, 80
    # registers -- than the counting ones above; what bench.py times): every byte, every sum and the totals must equal the
    # frame the oracle windows just pinned.  For C4 with the chunk forced to 64 spp as well (default: 79 per pass).
    for chunk in ((0, 64) if name == "gpu_showcase" else (0,)):
        img2 = np.zeros((h, w, 4), np.uint8)
        acc2 = np.zeros((h, w, 3))
        st2 = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, chunk), img2, None, acc2, ctx=gpu_ctx)
        assert np.array_equal(acc, acc2) and np.array_equal(img, img2)
        assert [st2[k] for k in ("samples", "segments", "exit_scans", "draws")] == [st[k] for k in ("samples", "segments", "exit_scans", "draws")]


def test_job_buffers_shrink_when_the_device_is_short_of_memory():
    # a job-buffer budget beyond the card's HBM: the library halves the samples per pass until the
    # buffers fit, and the pixels do not depend on that
    import os

    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path("test_scene"))
    w, h, spp, depth = 1920, 1080, 4096, 2
    imgs = []
    for budget in ("1000000", None):  # 1 TB of job buffers asked for, then the default
        old = os.environ.pop("PTCORE_L_BUDGET_MB", None)
        if budget:
            os.environ["PTCORE_L_BUDGET_MB"] = budget
        try:
            with capi.Context(ndev=1) as ctx:
                img = np.zeros((h, w, 4), np.uint8)
                st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 3), img, ctx=ctx)
                imgs.append((img, st["spp_chunk"], st["segments"]))
        finally:
            os.environ.pop("PTCORE_L_BUDGET_MB", None)
            if old is not None:
                os.environ["PTCORE_L_BUDGET_MB"] = old
    assert imgs[0][1] < spp  # 4096 spp per pass would need 760 GB; it was reduced
    assert np.array_equal(imgs[0][0], imgs[1][0]) and imgs[0][2] == imgs[1][2]


@pytest.mark.parametrize("w,h,spp", [(33, 33, 4096), (400, 225, 2048)])
def test_many_samples_on_ragged_frames(gpu_ctx, oracle, w, h, spp):
    # Edge tiles hold long runs of jobs whose pixel lies outside the frame (256*S in a row); with thousands of
    # samples per pass there are more such claims than resident waves.  Every in-frame job must still be traced.
    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path("test_scene"))
    depth, seed = 2, 5
    o = oracle.render(oracle.Scene.load(scene_path("test_scene")), w, h, spp, depth, seed=seed)
    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3))
    nseg = np.zeros((h, w), np.uint32)
    ndraw = np.zeros((h, w), np.uint32)
    st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw,
                    ctx=gpu_ctx)
    assert st["spp_chunk"] >= 1024  # thousands of samples in one pass
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"]
    assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"])
    assert np.array_equal(img, o["rgba"])
    img2 = np.zeros((h, w, 4), np.uint8)
    acc2 = np.zeros((h, w, 3))
    st2 = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed), img2, None, acc2, ctx=gpu_ctx)  # the build that ships
    assert st2["spp_chunk"] >= 1024 and st2["segments"] == o["stats"]["segments"] and st2["draws"] == o["stats"]["draws"]
    assert np.array_equal(img2, o["rgba"]) and np.array_equal(acc2, acc)
