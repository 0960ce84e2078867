"""rt_photons_build's GPU kd-tree against the oracle's BalanceSegment (src/photonmap.h:176-264) on synthetic photon
sets: every node of the heap must be the same photon with the same plane and scaled power, for sizes around the
points where the build changes method (one lane, one wave, one workgroup per segment), for coordinates drawn from a
handful of values (all photons of a wall share one) and for sorted input."""
import numpy as np
import pytest

import oraclelib as O
from raytracer_amd import abi, lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = lib.Context(0)
    c.upload(O.scene_from_golden(O.load_golden("scene_a")).desc)
    yield c
    c.close()


def photons(n, kind, seed):
    rng = np.random.default_rng(seed)
    p = np.zeros(n, dtype=abi.PHOTON_DTYPE)
    if kind == "uniform":
        pos = rng.uniform(-20, 30, size=(n, 3))
    elif kind == "ties":       # few distinct coordinates: the order of equal keys decides the tree
        pos = rng.integers(-3, 4, size=(n, 3)) * 1.5
    elif kind == "walls":      # photons on the faces of a box
        pos = rng.uniform(-10, 10, size=(n, 3))
        face = rng.integers(0, 6, size=n)
        pos[np.arange(n), face % 3] = np.where(face < 3, -10.0, 10.0)
    elif kind == "sorted":
        pos = np.sort(rng.uniform(1, 2, size=(n, 3)), axis=0)
    p["pos"] = pos.astype(np.float32)
    p["power"] = rng.uniform(0.1, 2.0, size=n).astype(np.float32)
    p["dir"] = rng.normal(size=(n, 3)).astype(np.float32)
    p["rgb"] = rng.integers(0, 256, size=(n, 3)).astype(np.uint8)
    return p


SIZES = [1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 63, 64, 65, 100, 1023, 1024, 1025, 4097, 33000, 70001]


@pytest.mark.parametrize("kind", ["uniform", "ties", "walls", "sorted"])
def test_tree_equals_reference_tree(ctx, kind):
    for n in SIZES:
        raw = photons(n, kind, n)
        want = O.photon_build(raw)
        ctx.photons_build(abi.RT_MAP_GLOBAL, raw)
        got = ctx.photons_download(abi.RT_MAP_GLOBAL)
        assert got.tobytes() == want.tobytes(), f"{kind} n={n}: first differing node {int(np.flatnonzero(got.view('V32') != want.view('V32'))[0]) + 1}"


def test_large_tree_equals_reference_tree(ctx):
    raw = photons(1_500_000, "walls", 3)
    want = O.photon_build(raw)
    ctx.photons_build(abi.RT_MAP_CAUSTICS, raw)
    got = ctx.photons_download(abi.RT_MAP_CAUSTICS)
    assert got.tobytes() == want.tobytes()


def test_non_finite_position_is_refused(ctx):
    raw = photons(100, "uniform", 1)
    raw["pos"][17, 1] = np.nan
    with pytest.raises(lib.RtError):
        ctx.photons_build(abi.RT_MAP_GLOBAL, raw)


def test_build_starts_over_without_shared_segments(ctx, monkeypatch):
    """Workgroups that share a segment wait for each other at barriers of their own; if they cannot all run at once
    (something else holds compute units) they give up, leaving the segment half partitioned.  rt_photons_build then
    starts over from the caller's photons with one workgroup per segment - here forced by the test hook, from the
    arrays a first complete build left behind: the same tree, node for node."""
    raw = photons(400_000, "walls", 11)
    want = O.photon_build(raw)
    ctx.photons_build(abi.RT_MAP_GLOBAL, raw)
    fast_ms = ctx.counters().kernel_ms
    assert ctx.photons_download(abi.RT_MAP_GLOBAL).tobytes() == want.tobytes()
    monkeypatch.setenv("RT_KD_TEST_ABORT", "1")
    ctx.photons_build(abi.RT_MAP_GLOBAL, raw)
    slow_ms = ctx.counters().kernel_ms
    monkeypatch.delenv("RT_KD_TEST_ABORT")
    assert ctx.photons_download(abi.RT_MAP_GLOBAL).tobytes() == want.tobytes()
    assert slow_ms > fast_ms  # both attempts are in the time
