"""
wrt_render_static, the entry point that allocates nothing (its path records live in a static
device array of the library), must give the accumulation image of wrt_render bit for bit: called
through the C ABI twice in a row, the second time with a smaller frame, so that what the first call
left in the shared array would show if anything depended on it.

It must also live up to its name: the free device memory is the same before and after the calls,
and a call under HIP stream capture (global mode, where an allocation or a synchronising call is an
error) records a graph whose replays give the same image.
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
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(ctx.torch_device)[0]       # hipMemGetInfo
        ctx.render_static(w, h, s_begin, s_end, bounces, seed, got)
        torch.cuda.synchronize()
        # no device memory taken or given back by the call (its workspace is the library's static array)
        assert torch.cuda.mem_get_info(ctx.torch_device)[0] == free_before
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


def test_static_workspace_render_under_stream_capture():
    """wrt_render_static called while its stream is being captured (global mode: hipMalloc, hipFree and
    synchronising calls would fail the capture) records the launches of one call; the graph's replays
    add the same samples to accum as wrt_render does, and afterwards the context still takes a new
    scene (its bookkeeping event was not drawn into the capture)."""
    import torch
    from tests import helpers
    from weekend_raytracer import _native
    demo = helpers.golden_scene(helpers.golden("frame_demo"))
    ctx = _native.Context(0)
    ctx.upload(demo)
    w, h, bounces, seed = 320, 180, 5, 77
    want = ctx.zeros((w, h, 3), torch.float32)
    ctx.render(w, h, 0, 4, bounces, seed, want)
    ctx.render(w, h, 0, 4, bounces, seed, want)                 # accum is added to: two calls, twice the samples
    got = ctx.zeros((w, h, 3), torch.float32)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="global"):
        # (inside, torch's current stream is the capturing one; nothing runs yet)
        ctx.render_static(w, h, 0, 4, bounces, seed, got)
    torch.cuda.synchronize()
    assert not got.cpu().numpy().any()
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy()) and want.cpu().numpy().any()
    # the context carries on: another scene, another render through both entry points
    ctx.upload(helpers.golden_scene(helpers.golden("frame_disks")))
    a, b = ctx.zeros((64, 36, 3), torch.float32), ctx.zeros((64, 36, 3), torch.float32)
    ctx.render(64, 36, 0, 2, bounces, seed, a)
    ctx.render_static(64, 36, 0, 2, bounces, seed, b)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()) and a.cpu().numpy().any()
    del graph
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
