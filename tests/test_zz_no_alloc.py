"""rt_trace_brief_noalloc (the entry point that allocates nothing) against rt_trace_brief, through the C ABI.

Its work queue, counters and the stack levels beyond LDS live in one static device array per GPU, which keeps the
previous call's contents: two calls in a row, the second with fewer rays, must each give exactly the answers of
rt_trace_brief, for closest-hit and any-hit queries, with the default LDS split and with three LDS levels (then the
static array holds nearly the whole stack)."""
import numpy as np
import pytest
import torch

import oraclelib as O
from raytracer_amd import abi, lib

pytestmark = pytest.mark.gpu


def brief_words(a):
    return np.ascontiguousarray(a).view(np.uint32).reshape(len(a), -1)


def run_noalloc(ctx, rays, tmax):
    d_rays = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
    d_tmax = torch.from_numpy(np.ascontiguousarray(tmax, np.float32)).cuda() if tmax is not None else None
    d_out = torch.full((len(rays), abi.HIT_BRIEF_DTYPE.itemsize), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    ctx.trace_brief_noalloc(d_rays.data_ptr(), d_tmax.data_ptr() if d_tmax is not None else 0, len(rays), d_out.data_ptr())
    assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the device's free memory: it allocated"
    return d_out.cpu().numpy().view(abi.HIT_BRIEF_DTYPE).reshape(-1)


@pytest.mark.parametrize("lds_levels", [None, 3])
def test_noalloc_equals_trace_brief_twice_in_a_row(lds_levels):
    mp = pytest.MonkeyPatch()
    if lds_levels:
        mp.setenv("RT_LDS_LEVELS", str(lds_levels))
    ctx = lib.Context(0)
    mp.undo()
    try:
        for name in ("kat_trace_mesh", "kat_trace_nested"):
            g = O.load_golden(name)
            ctx.upload(O.scene_from_golden(g).desc)
            rays, tmax = np.tile(g["kat_rays"], (9, 1)), np.tile(g["kat_tmax"], 9)
            for n in (len(rays), 1000, 1):  # each call after the first finds the previous call's queue word and stacks
                want = ctx.trace_brief(rays[:n])
                got = run_noalloc(ctx, rays[:n], None)
                assert np.array_equal(brief_words(got), brief_words(want)), (name, n, "closest")
                assert np.array_equal(got["node"], np.tile(g["kat_id"], 9)[:n])  # and the reference's own answers
                want = ctx.trace_brief(rays[:n], tmax[:n])
                got = run_noalloc(ctx, rays[:n], tmax[:n])
                assert np.array_equal(brief_words(got), brief_words(want)), (name, n, "any-hit")
                assert np.array_equal(got["node"], np.tile(g["kat_flags"][:, 2], 9)[:n])
                assert ctx.counters().kernel_ms > 0
        ctx.trace_brief_noalloc(0, 0, 0, 0)  # no rays: nothing to do
        with pytest.raises(lib.RtError) as e:
            ctx.trace_brief_noalloc(1, 0, abi.RT_NOALLOC_MAX_RAYS + 1, 1)
        assert e.value.code == abi.RT_ERR_INVALID and "at most" in str(e.value)
        with pytest.raises(lib.RtError):
            ctx.trace_brief_noalloc(0, 0, 5, 0)
    finally:
        ctx.close()
