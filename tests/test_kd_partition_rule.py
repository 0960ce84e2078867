"""The rank rule the GPU kd-tree build uses for one partition step of the quick-select of
PhotonMap::BalanceSegment (src/photonmap.h:221-238), checked against the two-pointer loops themselves.

The device code (raytracer_amd/csrc/device/kdbuild.h) finds, for every position of [left, right], whether the
step swaps it and with whom from two prefix sums; this file states the same rule with numpy and compares the
array it leaves, and the place the pivot lands, with the serial step on arrays full of equal keys."""
import numpy as np


def serial_step(key, left, right):
    key = key.copy()
    v = key[right]
    i, j = left - 1, right
    while True:
        i += 1
        if not key[i] < v: break
    while True:
        j -= 1
        if not (key[j] > v and j > left): break
    while i < j:
        key[i], key[j] = key[j], key[i]
        while True:
            i += 1
            if not key[i] < v: break
        while True:
            j -= 1
            if not (key[j] > v and j > left): break
    key[i], key[right] = key[right], key[i]
    return key, i


def rank_step(key, left, right):
    key = key.copy()
    v = key[right]
    pos = np.arange(left, right + 1)
    k = key[left:right + 1]
    f_l = k >= v
    f_r = (k <= v) | (pos == left)
    f_r[-1] = False                                   # j starts below `right`
    c_l = np.cumsum(f_l) - f_l                        # L positions before p
    r_r = f_r.sum() - np.cumsum(f_r)                  # R positions after p
    swap_l = f_l & (r_r > c_l)
    swap_r = f_r & (c_l > r_r)
    assert not (swap_l & swap_r).any()
    lpos = np.empty(swap_l.sum(), dtype=np.int64); lpos[c_l[swap_l]] = pos[swap_l]
    rpos = np.empty(swap_r.sum(), dtype=np.int64); rpos[r_r[swap_r]] = pos[swap_r]
    assert len(lpos) == len(rpos)
    i_final = pos[f_l & ~swap_l][0]                  # first L position the step does not swap ...
    assert c_l[i_final - left] == len(lpos)
    if len(rpos): i_final = min(i_final, rpos[-1])    # ... unless i meets the key its last swap moved up first
    key[lpos], key[rpos] = key[rpos].copy(), key[lpos].copy()
    key[i_final], key[right] = key[right], key[i_final]
    return key, i_final


def test_rank_rule_equals_two_pointer_step():
    rng = np.random.default_rng(5)
    for trial in range(3000):
        n = int(rng.integers(2, 40))
        distinct = int(rng.integers(1, 6)) if trial % 2 else 1000
        key = rng.integers(0, distinct, size=n + 3).astype(np.float32)
        left = int(rng.integers(1, 3)); right = left + n - 1
        a, ia = serial_step(key, left, right)
        b, ib = rank_step(key, left, right)
        assert ia == ib, (trial, key, left, right)
        assert np.array_equal(a, b), (trial, key, left, right)
