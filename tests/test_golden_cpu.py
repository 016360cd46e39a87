"""Oracle vs the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENE_NAMES, scene_path


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_oracle_reproduces_golden(oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    w, h, spp, depth, seed = (int(x) for x in g["cfg"])
    r = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=seed)
    assert np.array_equal(r["rgba"], g["rgba"])
    assert np.array_equal(r["accum"], g["accum"])  # same machine arithmetic: bit-equal
    assert np.array_equal(r["nseg"], g["nseg"]) and np.array_equal(r["ndraw"], g["ndraw"])
    st = r["stats"]
    assert [st["samples"], st["segments"], st["exit_scans"], st["draws"]] == [int(x) for x in g["totals"]]


def test_two_seeds_differ_by_monte_carlo_noise_only(oracle):
    # two seeds at N spp differ by about sqrt(2)*sigma/sqrt(N), not by a bias
    sc = oracle.Scene.load(scene_path("example_simple"))
    w, h = 48, 27
    a8 = oracle.render(sc, w, h, 8, 6, seed=1, want=("accum",))["accum"] / 8
    b8 = oracle.render(sc, w, h, 8, 6, seed=2, want=("accum",))["accum"] / 8
    a64 = oracle.render(sc, w, h, 64, 6, seed=1, want=("accum",))["accum"] / 64
    b64 = oracle.render(sc, w, h, 64, 6, seed=2, want=("accum",))["accum"] / 64
    d8 = np.sqrt(np.mean((a8 - b8) ** 2))
    d64 = np.sqrt(np.mean((a64 - b64) ** 2))
    assert d8 > 0 and d64 > 0
    ratio = d8 / d64
    assert 1.8 < ratio < 4.5, ratio  # ideal sqrt(64/8) = 2.83
    assert abs(np.mean(a64) - np.mean(b64)) < 0.05 * np.mean(a64)  # no seed-dependent bias in the mean
