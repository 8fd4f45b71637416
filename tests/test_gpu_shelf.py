"""
The path shelf (wrt_set_path_shelf: a second path per lane of the bounce kernel, kept in device memory)
must never change a result: a path is the same whichever lane traces it and whenever.  Frames with the
shelf forced on and switched off are compared path by path, bit for bit, and with the fixtures the
reference produced; among them scenes with one, two, three and four large groups, a frame whose last
batches are ragged, and a frame of a single sample (the shelf's paths outlive the work counter: they are
finished after the last batch has been handed out).
"""
import contextlib
import io

import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_parity import frame_parity

pytestmark = pytest.mark.gpu


def paths_with_shelf(objs, w, h, s, b, seed, mode, s_begin=0):
    import torch
    from weekend_raytracer import main
    frame = main.Frame(objs, hierarchy=0)
    frame.ctx.set_path_shelf(mode)
    frame.ctx.stats_reset()
    paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
    accum = frame.accumulate(w, h, s_begin, s_begin + s, b, seed, paths=paths)
    stats = frame.ctx.stats()
    out = paths.cpu().numpy(), accum.cpu().numpy(), stats
    frame.ctx.close()
    return out


def scene_objects(name, aspect):
    from weekend_raytracer import scenes
    with contextlib.redirect_stdout(io.StringIO()):
        return getattr(scenes, name)(aspect)


@pytest.mark.parametrize("name,size", [
    ("ray_group_triangle_group_bunny_scene", (320, 180, 6, 10)),      # one large group
    ("cloth_bunny_scene", (250, 141, 4, 8)),                          # two, ragged tiles
    ("cow_teapot_field_scene", (160, 90, 2, 6)),                      # two meshes and a sphere field
    ("cow_teapot_field_scene", (333, 187, 1, 10)),
    ("numpy_bunnies_scene", (160, 90, 3, 10)),                        # four meshes
    # frames of several paths per lane of the resident grid: every wavefront refills many times
    ("cow_teapot_field_scene", (960, 540, 4, 10)),
    ("numpy_bunnies_scene", (1280, 720, 3, 10)),
    ("cloth_bunny_scene", (1001, 563, 5, 10)),
])
def test_shelf_changes_nothing(name, size):
    w, h, s, b = size
    objs = scene_objects(name, w / h)
    p_on, a_on, st_on = paths_with_shelf(objs, w, h, s, b, 77, 1)
    p_off, a_off, st_off = paths_with_shelf(objs, w, h, s, b, 77, 0)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert np.array_equal(a_on.view(np.uint32), a_off.view(np.uint32))
    assert np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"])
    # the same pairs are looked at either way: only which lanes look at them, and when, differs
    for key in ("triangle_pairs", "sphere_pairs", "cull_tests"):
        assert st_on[key] == st_off[key], key
    assert st_on["shelved_paths"] > 0 and st_off["shelved_paths"] == 0


@pytest.mark.parametrize("name", ["cow_teapot", "bunnies", "cloth_bunny", "bunny_textured", "triangles"])
def test_fixture_frames_with_the_shelf(name, monkeypatch):
    monkeypatch.setenv("WRT_PATH_SHELF", "1")
    f, r = frame_parity(name)
    assert np.array_equal(r["path_colours"], f["path_colours"])
    assert np.array_equal(r["segments"], f["segments"])
    assert np.array_equal(r["counts"], f["counts"])
    assert np.array_equal(r["image"], f["image"], equal_nan=True)


def test_automatic_mode_picks_scenes_with_two_meshes():
    """Default mode: the shelf runs where two expensive loops compete (cow + teapot) and nowhere else."""
    w, h, s, b = 512, 288, 4, 8
    objs = scene_objects("cow_teapot_field_scene", w / h)
    _, _, auto = paths_with_shelf(objs, w, h, s, b, 5, -1)
    _, _, off = paths_with_shelf(objs, w, h, s, b, 5, 0)
    assert auto["triangle_pairs"] == off["triangle_pairs"]
    assert auto["shelved_paths"] > 0 and off["shelved_paths"] == 0
    objs = scene_objects("ray_group_triangle_group_bunny_scene", w / h)
    _, _, auto = paths_with_shelf(objs, w, h, s, b, 5, -1)
    _, _, on = paths_with_shelf(objs, w, h, s, b, 5, 1)
    assert auto["shelved_paths"] == 0 and on["shelved_paths"] > 0
