"""Oracle KATs for the post-process passes of the reference GPU backend (gpu.go:22-47, :2309-2520)."""
import numpy as np


def test_aces_curve_points(oracle):
    L = oracle.lib()
    assert L.ora_aces_tonemap(0.0) == 0.0 and L.ora_aces_tonemap(-3.0) == 0.0
    x = 1.0
    want = np.float32((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14))
    assert L.ora_aces_tonemap(1.0) == float(want)
    assert L.ora_aces_tonemap(1e6) == 1.0  # clamped
    vals = [L.ora_aces_tonemap(float(v)) for v in np.linspace(0, 8, 200, dtype=np.float32)]
    assert all(b >= a for a, b in zip(vals, vals[1:]))  # monotone


def test_tonemap_rounds_to_nearest_in_float32(oracle):
    acc = np.zeros((3, 3, 3))
    acc[1, 1] = [4.0, 0.18 * 2, 1e9]
    img = np.zeros((3, 3, 4), np.uint8)
    oracle.post_process(img, tonemap=True, accum=acc, spp=2)
    assert np.all(img[..., 3] == 255) and np.all(img[0, 0, :3] == 0)
    g = np.sqrt(np.float32(oracle.lib().ora_aces_tonemap(2.0)))
    assert img[1, 1, 0] == int(np.float32(np.float32(g) * np.float32(255.0)) + np.float32(0.5))
    assert img[1, 1, 2] == 255


def test_bilateral_keeps_flat_images_and_edges(oracle):
    img = np.full((8, 9, 4), 77, np.uint8)
    img[..., 3] = 255
    ref = img.copy()
    oracle.post_process(img, denoise=True)
    assert np.array_equal(img, ref)  # constant image is a fixed point
    edge = np.zeros((8, 8, 4), np.uint8)
    edge[..., 3] = 255
    edge[:, 4:, :3] = 255
    out = edge.copy()
    oracle.post_process(out, denoise=True, sigma_r=0.05)
    assert np.array_equal(out, edge)  # a hard edge survives a small range sigma
    tiny = np.arange(2 * 2 * 4, dtype=np.uint8).reshape(2, 2, 4)
    t2 = tiny.copy()
    oracle.post_process(t2, denoise=True, smooth=True)
    assert np.array_equal(t2, tiny)  # skipped unless w > 2 and h > 2 (gpu.go:2358, :2447)


def test_box_blur_on_an_impulse(oracle):
    img = np.zeros((7, 7, 4), np.uint8)
    img[..., 3] = 255
    img[3, 3, :3] = 250
    out = img.copy()
    oracle.post_process(out, smooth=True, smooth_radius=1, smooth_strength=1.0)
    assert out[3, 3, 0] == int(250 / 9 + 0.5) and out[2, 2, 0] == int(250 / 9 + 0.5) and out[1, 1, 0] == 0
    assert out[0, 0, 0] == 0 and np.all(out[..., 3] == 255)
    half = img.copy()
    oracle.post_process(half, smooth=True, smooth_radius=9, smooth_strength=0.5)  # radius clamps to 5: whole 7x7 window
    assert half[3, 3, 0] == int(0.5 * 250 + 0.5 * (250 / 49) + 0.5)
