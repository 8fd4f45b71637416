"""
wrt_render_static, the entry point that allocates nothing (its path records live in a static
device array of the library), must give the accumulation image of wrt_render bit for bit: called
through the C ABI twice in a row, the second time with a smaller frame, so that what the first call
left in the shared array would show if anything depended on it.
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_workspace_render_equals_wrt_render():
    import torch
    from weekend_raytracer import _native, main, scenes
    with contextlib.redirect_stdout(io.StringIO()):
        objs = scenes.ray_group_triangle_group_bunny_scene(16 / 9)
    frame = main.Frame(objs)
    ctx = frame.ctx
    seed, bounces = 4242, 6
    # first call: several launches (the samples of 1600x900 do not fit the array at once); second: smaller
    for (w, h, s_begin, s_end) in ((1600, 900, 1, 6), (200, 120, 0, 3)):
        want = ctx.zeros((w, h, 3), torch.float32)
        ctx.stats_reset()
        ctx.render(w, h, s_begin, s_end, bounces, seed, want)
        stats_want = ctx.stats()
        got = ctx.zeros((w, h, 3), torch.float32)
        ctx.stats_reset()
        ctx.render_static(w, h, s_begin, s_end, bounces, seed, got)
        stats_got = ctx.stats()
        a, b = want.cpu().numpy(), got.cpu().numpy()
        assert a.any()
        assert np.array_equal(a, b), (w, h)                       # bit-exact: same kernels, same order of sums
        assert np.array_equal(stats_want["rays_per_bounce"], stats_got["rays_per_bounce"])
        assert stats_want["triangle_pairs"] == stats_got["triangle_pairs"]
    # it adds into accum like wrt_render does
    ctx.render_static(200, 120, 3, 5, bounces, seed, got)
    ctx.render(200, 120, 3, 5, bounces, seed, want)
    assert np.array_equal(want.cpu().numpy(), got.cpu().numpy())
    # beyond the array: a clear error, nothing launched
    too_wide = ctx.zeros((2, 2, 3), torch.float32)
    with pytest.raises(_native.NativeError, match="WRT_STATIC_MAX_PIXELS"):
        ctx.render_static(4096, 1025, 0, 1, bounces, seed, too_wide)
    ctx.close()


def test_static_workspace_against_the_oracle():
    """The same entry point against the CPU oracle, every path's sum per pixel."""
    import torch
    from oracle import numpy_path
    from weekend_raytracer import _flatten, main, scenes
    with contextlib.redirect_stdout(io.StringIO()):
        objs = scenes.numpy_one_weekend_demo_scene(48 / 27)
    flat = _flatten.flatten(*objs, main.HORIZON_COLOUR, main.SKY_COLOUR)
    frame = main.Frame(objs)
    w, h, s, b, seed = 48, 27, 3, 5, 99
    accum = frame.ctx.zeros((w, h, 3), torch.float32)
    frame.ctx.render_static(w, h, 0, s, b, seed, accum)
    want = numpy_path.render(flat, w, h, s, b, seed)
    assert np.array_equal(accum.cpu().numpy(), want["sample_sum"])
    frame.ctx.close()
