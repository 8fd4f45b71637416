"""
Parity at the benchmark configurations' full sizes.  The CPU oracle cannot
render 60-90 million paths, but the random streams are keyed per path, so any
subset of paths can be re-traced on the CPU: a random sample of the GPU's
paths, plus every path whose colour leaves [0, 1], is compared with the oracle.
Size-independent properties are checked on the whole frame.
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import numpy_path
from tests import helpers

pytestmark = pytest.mark.gpu


def full_frame(scene_fn, w, h, s, b, seed, hierarchy=None):
    import torch
    from weekend_raytracer import _flatten, main
    with contextlib.redirect_stdout(io.StringIO()):
        objs = scene_fn(w / h)
    flat = _flatten.flatten(*objs, main.HORIZON_COLOUR, main.SKY_COLOUR)
    frame = main.Frame(objs, hierarchy=hierarchy)
    frame.ctx.stats_reset()
    paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
    accum = frame.accumulate(w, h, 0, s, b, seed, paths=paths)
    stats = frame.ctx.stats()
    p = paths.cpu().numpy()
    a = accum.cpu().numpy()
    frame.ctx.close()
    return flat, p, a, stats


def compare_subset(flat, p, sel, w, h, s, b, seed):
    pix, smp = np.divmod(sel, s)
    x, y = np.divmod(pix, h)
    r = numpy_path.trace_paths(flat, w, h, x, y, smp, b, seed)
    same = np.all(r["path_colours"] == p[sel, :3], axis=1)
    segs = (p[sel, 3].astype(np.int32) % 100) == r["segments"]
    return same, segs


def check_full_frame(flat, p, a, stats, w, h, s, b, seed, n_groups_pairs, sampled=4000):
    n = w * h * s
    # every path written exactly once, by construction of the index
    assert not np.any(p[:, 3] == -7.0)
    segs = p[:, 3].astype(np.int32)
    survivors = segs >= 100
    segs = segs % 100
    assert segs.min() >= 1 and segs.max() <= b
    assert np.all(p[survivors, :3] == 0.0) and np.all(segs[survivors] == b)
    # the statistics are the histogram of what was written
    counts = stats["rays_per_bounce"]
    assert stats["paths"] == n and counts[0] == n
    for k in range(1, b):
        assert counts[k] == int(np.sum(segs > k))
    assert counts[b] == int(survivors.sum())
    assert np.all(np.diff(counts[:b + 1]) <= 0)
    # the accumulation image is the per-pixel sum of its paths (float32, sample order)
    ref = np.zeros((w * h, 3), dtype=np.single)
    pp = p[:, :3].reshape(w * h, s, 3)
    for k in range(s):
        ref += pp[:, k]
    assert np.array_equal(ref.reshape(w, h, 3), a)
    # sampled paths against the CPU oracle
    rng = np.random.default_rng(2025)
    sel = np.sort(rng.choice(n, size=sampled, replace=False))
    same, seg_ok = compare_subset(flat, p, sel, w, h, s, b, seed)
    assert same.all(), same.mean()
    assert seg_ok.all()
    # paths the unnormalised refraction of the reference blows up (colour outside [0,1])
    odd = np.where((p[:, :3].min(axis=1) < 0) | (p[:, :3].max(axis=1) > 1.0001))[0][:300]
    if odd.size:
        same_odd, _ = compare_subset(flat, p, odd, w, h, s, b, seed)
        assert same_odd.all(), (odd.size, same_odd.mean())
    return float(same.mean())


def test_cover_scene_full_size():
    """The sphere benchmark configuration: 484 spheres, depth of field, 1280x720, 100 spp."""
    from weekend_raytracer import scenes
    w, h, s, b, seed = 1280, 720, 100, 10, 1234
    flat, p, a, stats = full_frame(lambda ar: scenes.cover_scene(ar, scenes.sphere_field_records(480)),
                                   w, h, s, b, seed)
    check_full_frame(flat, p, a, stats, w, h, s, b, seed, None)
    assert stats["sphere_pairs"] == int(stats["rays_per_bounce"][:b].sum()) * 484


def test_bunny_scene_full_size():
    """The mesh benchmark configuration: bunny.obj, every ray against every triangle, smooth normals, 1920x1080, 30 spp, 10 bounces."""
    from weekend_raytracer import scenes
    w, h, s, b, seed = 1920, 1080, 30, 10, 1234
    flat, p, a, stats = full_frame(scenes.bunny_diffuse_scene, w, h, s, b, seed)
    check_full_frame(flat, p, a, stats, w, h, s, b, seed, None)
    segments = int(stats["rays_per_bounce"][:b].sum())
    assert stats["cull_tests"] == segments and stats["sphere_pairs"] == segments
    assert stats["triangle_pairs"] % 644 == 0 and 0 < stats["triangle_pairs"] <= segments * 644


@pytest.mark.parametrize("hierarchy", [False, 64])
def test_cloth_bunny_scene_full_size(hierarchy):
    """The textured configuration: textured bunny + UV-mapped picnic cloth, 1920x1080, 64 spp, 10 bounces (the
    reference has no strength maps, so none are applied); every pair examined and, apart, both meshes walked
    through their hierarchies."""
    from weekend_raytracer import scenes
    w, h, s, b, seed = 1920, 1080, 64, 10, 1234
    flat, p, a, stats = full_frame(scenes.cloth_bunny_scene, w, h, s, b, seed, hierarchy=hierarchy)
    check_full_frame(flat, p, a, stats, w, h, s, b, seed, None, sampled=3000)
    assert 0 < stats["triangle_evaluations"] < stats["triangle_pairs"]


@pytest.mark.parametrize("hierarchy", [False, True])
def test_cow_teapot_field_full_size(hierarchy):
    """The multi-GPU benchmark configuration's per-GPU workload: cow.obj + teapot.obj (12 124 triangles) + the
    sphere field at 3840x2160 (8 of its 128 samples per pixel: the streams are keyed per sample, so any
    sample range checks the same code), every pair examined and, apart, with the optional hierarchy."""
    from weekend_raytracer import scenes
    w, h, s, b, seed = 3840, 2160, 8, 10, 1234
    flat, p, a, stats = full_frame(scenes.cow_teapot_field_scene, w, h, s, b, seed, hierarchy=hierarchy)
    check_full_frame(flat, p, a, stats, w, h, s, b, seed, None, sampled=2000)
    segments = int(stats["rays_per_bounce"][:b].sum())
    assert stats["sphere_pairs"] == segments * 191 and stats["cull_tests"] == 2 * segments
    assert 0 < stats["triangle_evaluations"] < stats["triangle_pairs"] <= segments * 12124


@pytest.mark.parametrize("scene_name,size", [
    ("numpy_one_weekend_demo_scene", (320, 180, 10, 5)),         # the reference's own CPU-runnable configuration
    ("numpy_glass_experiment_scene", (320, 180, 6, 12)),         # nested glass: long refraction chains
])
def test_sphere_configurations_every_path(scene_name, size):
    """Small enough for the CPU oracle to trace in full: every path, count and pixel must be equal."""
    from weekend_raytracer import _flatten, main, scenes
    w, h, s, b = size
    with contextlib.redirect_stdout(io.StringIO()):
        objs = getattr(scenes, scene_name)(w / h)
    got = main.render_frame(objs, w, h, s, b, seed=4321, want_paths=True)
    flat = _flatten.flatten(*objs, main.HORIZON_COLOUR, main.SKY_COLOUR)
    want = numpy_path.render(flat, w, h, s, b, 4321)
    assert np.array_equal(got["path_colours"], want["path_colours"])
    assert np.array_equal(got["segments"], want["segments"])
    assert np.array_equal(got["counts"], want["counts"])
    assert np.array_equal(got["image"], want["image"], equal_nan=True)
