"""
The reference's own class-level interface, driven the way reference main.py drives it:
Camera.get_ray_components (camera.py:108), <group>.get_hits (sphere_group.py:72,
triangle_group.py:150, disk.py:48), <material>.scatter (materials.py:61,259,314), main.render()
(main.py:77,223-232) and generate_image_from_data (main.py:45-74).

The bounce loop below is this repository's own restatement of main.py:110-212 and uses nothing but
those three calls; with the keyed streams named through camera.set_stream / materials.set_stream it
must reproduce, path by path, the frames the reference produced under the same streams
(tests/golden/frame_*.npz).
"""
import contextlib
import io

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def per_object_bounce_loop(scene, width, height, samples, max_bounces, seed, s_begin=0, trace=None):
    from weekend_raytracer import camera as camera_module
    from weekend_raytracer import main, materials
    cam, groups, material_map = scene
    camera_module.set_stream(seed, s_begin)
    origins, dirs = cam.get_ray_components(width, height, samples)
    assert origins.shape == (width, height, samples, 3) and origins.dtype == np.single
    origins, dirs = origins.reshape(-1, 3).copy(), dirs.reshape(-1, 3).copy()
    n = origins.shape[0]
    ids = np.arange(n)
    px, py, ps = ids // (height * samples), (ids // samples) % height, ids % samples + s_begin
    colours = np.ones((n, max_bounces + 1, 3), dtype=np.single)
    segments = np.zeros(n, dtype=np.int32)
    active = np.arange(n)
    counts = []
    for bounce in range(max_bounces + 1):
        counts.append(active.shape[0])
        if bounce == max_bounces:
            colours[active, bounce] = 0.0
            break
        segments[active] += 1
        m = active.shape[0]
        hit = np.full(m, False)
        best_t = np.full(m, 5001.0)
        pts, normals, uvs = np.zeros((m, 3)), np.zeros((m, 3)), np.zeros((m, 2))
        mats = np.full(m, 255, dtype=np.ubyte)
        back = np.full(m, False)
        for group in groups:
            g_hit, g_t, g_p, g_n, g_uv, g_mat, g_back = group.get_hits(origins[active], dirs[active], 0.001, 1000.0)
            assert g_hit.shape == (m,) and g_p.shape == (m, 3) and g_uv.shape == (m, 2)
            take = g_hit & (g_t < best_t)
            hit |= g_hit
            best_t = np.where(take, g_t, best_t)
            pts = np.where(take[:, None], g_p, pts)
            normals = np.where(take[:, None], g_n, normals)
            uvs = np.where(take[:, None], g_uv, uvs)
            mats = np.where(take, g_mat, mats)
            back = np.where(take, g_back, back)
        absorbed = np.full(m, False)
        for index, material in material_map.items():
            match = hit & (mats == index)
            sel = active[match]
            materials.set_stream(seed, px[sel], py[sel], ps[sel], bounce)
            new_o, new_d, cols, gone = material.scatter(dirs[sel], pts[match], normals[match], uvs[match], back[match])
            origins[sel] = new_o
            dirs[sel] = new_d
            colours[sel, bounce] = cols
            absorbed[match] = gone
        missed = active[~hit]
        ts = (dirs[missed, 1] + 1.0) * 0.5
        colours[missed, bounce] = (1.0 - ts)[:, None] * main.HORIZON_COLOUR + ts[:, None] * main.SKY_COLOUR
        if trace is not None:
            trace.append(dict(hit=hit, t=best_t, p=pts, n=normals, uv=uvs, mat=mats, back=back,
                              out_d=dirs[active].copy(), colour=colours[active, bounce].copy(), absorbed=absorbed))
        active = active[hit & ~absorbed]
    path_colours = np.prod(colours, axis=1)
    image = np.sqrt(np.mean(path_colours.reshape(width, height, samples, 3), axis=2))
    return dict(path_colours=path_colours, segments=segments, counts=np.array(counts, dtype=np.int64), image=image)


def scene_objects(name, aspect):
    from weekend_raytracer import scenes
    fn = {"demo": scenes.numpy_one_weekend_demo_scene, "bunny_textured": scenes.ray_group_triangle_group_bunny_scene,
          "disks": scenes.disk_test_scene, "demo_shard": scenes.numpy_one_weekend_demo_scene}[name]
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(aspect)


@pytest.mark.parametrize("name", ["demo", "demo_shard", "bunny_textured", "disks"])
def test_per_object_interface_reproduces_reference_frames(name):
    from weekend_raytracer import _native
    f = helpers.golden("frame_" + name)
    w, h, s, b, s_begin = (int(v) for v in f["size"])
    trace = []
    scene = scene_objects(name, w / h)
    _native.release_resident()
    uploads_before = _native.upload_count
    got = per_object_bounce_loop(scene, w, h, s, b, int(f["seed"]), s_begin, trace)
    # every group, every material and the camera go to the GPU once per frame, not once per bounce
    assert _native.upload_count - uploads_before <= len(scene[1]) + len(scene[2]) + 1
    same = np.all(got["path_colours"] == f["path_colours"], axis=1)
    assert same.all(), (name, same.mean())
    assert np.array_equal(got["segments"], f["segments"])
    assert np.array_equal(got["counts"], f["counts"])
    assert np.array_equal(got["image"], f["image"], equal_nan=True)
    if "trace_b0_hit" in f:
        # the fixtures with per-bounce traces: every intermediate the reference's loop held
        for bounce, rec in enumerate(trace):
            hit = f[f"trace_b{bounce}_hit"]
            assert np.array_equal(rec["hit"], hit), (name, bounce)
            for k in ("t", "p", "n", "uv", "mat", "back"):
                assert np.array_equal(rec[k][hit], f[f"trace_b{bounce}_{k}"][hit]), (name, bounce, k)
            assert np.array_equal(rec["out_d"], f[f"trace_b{bounce}_out_d"]), (name, bounce)
            assert np.array_equal(rec["colour"], f[f"trace_b{bounce}_colour"]), (name, bounce)
            assert np.array_equal(rec["absorbed"], f[f"trace_b{bounce}_absorbed"]), (name, bounce)


def test_main_render_and_image_output(tmp_path, monkeypatch):
    """main.render() with the module settings of the reference (main.py:16-22,79-91) -> {(x, y): rgb};
    generate_image_from_data -> int(255 v) per channel, row 0 at the top (main.py:57-65)."""
    from PIL import Image
    from weekend_raytracer import main, scenes
    f = helpers.golden("frame_demo")
    w, h, s, b, _ = (int(v) for v in f["size"])
    monkeypatch.setattr(main, "IMG_WIDTH", w)
    monkeypatch.setattr(main, "IMG_HEIGHT", h)
    monkeypatch.setattr(main, "ASPECT_RATIO", w / h)
    monkeypatch.setattr(main, "PIXEL_SAMPLES", s)
    monkeypatch.setattr(main, "MAX_BOUNCES", b)
    monkeypatch.setattr(main, "SEED", int(f["seed"]))
    monkeypatch.setattr(main, "SCENE", scenes.numpy_one_weekend_demo_scene)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        pixels = main.render()
    assert set(pixels) == {(x, y) for x in range(w) for y in range(h)}
    image = np.array([[pixels[(x, y)] for y in range(h)] for x in range(w)])
    assert image.dtype == np.single and np.array_equal(image, f["image"])
    # the reference prints the active ray count of every bounce (main.py:111)
    for bounce, rays in enumerate(f["counts"]):
        assert f"Bounce {bounce} of {b}, {int(rays)} rays." in out.getvalue()
    path = tmp_path / "frame.png"
    main.generate_image_from_data(pixels, path=str(path))
    stored = np.asarray(Image.open(path))
    assert stored.shape == (h, w, 3) and stored.dtype == np.uint8
    for (x, y) in ((0, 0), (w - 1, 0), (3, h - 1), (w // 2, h // 3)):
        want = [int(255 * v) for v in pixels[(x, y)]]                 # main.py:60-64, a float32 product
        assert list(stored[h - y - 1, x]) == want                      # main.py:57: y flipped
    whole = np.array([[[int(255 * v) for v in pixels[(x, h - row - 1)]] for x in range(w)] for row in range(h)])
    assert np.array_equal(stored, whole)
    # the device-side quantisation gives the same bytes
    assert np.array_equal(main.image_bytes_on_device(f["image"]), stored)


def test_mutated_objects_are_uploaded_again():
    """The reference's objects are mutable builders: a change between two calls must reach the GPU."""
    from weekend_raytracer import camera as camera_module
    from weekend_raytracer import materials
    from weekend_raytracer.camera import Camera
    from weekend_raytracer.sphere_group import SphereGroup
    n = 16
    dirs = np.tile(np.array([[0.0, -1.0, 0.0]], dtype=np.single), (n, 1))
    pts = np.zeros((n, 3))
    normals = np.tile(np.array([[0.0, 1.0, 0.0]]), (n, 1))
    uvs, back = np.zeros((n, 2)), np.zeros(n, dtype=bool)
    mat = materials.Diffuse(np.array([0.1, 0.2, 0.3]))
    materials.set_stream(5, np.arange(n), np.zeros(n, int), np.zeros(n, int), 0)
    first = mat.scatter(dirs, pts, normals, uvs, back)
    assert np.allclose(first[2], [0.1, 0.2, 0.3])
    mat.colour = np.array([0.9, 0.8, 0.7])
    second = mat.scatter(dirs, pts, normals, uvs, back)
    assert np.allclose(second[2], [0.9, 0.8, 0.7]) and np.array_equal(first[1], second[1])
    metal = materials.Metal(np.array([1.0, 1.0, 1.0]), 0.0)
    mirror = metal.scatter(dirs, pts, normals, uvs, back)[1]
    metal.fuzziness = 0.5
    assert not np.array_equal(metal.scatter(dirs, pts, normals, uvs, back)[1], mirror)
    # camera: the lens radius is read at every call
    cam = Camera(np.array([0.0, 0.0, 5.0]), np.array([0.0, 0.0, 0.0]), 5.0, 0.0, 2.0, 40.0)
    camera_module.set_stream(3)
    o0, _ = cam.get_ray_components(8, 4, 2)
    assert np.all(o0 == np.array([0.0, 0.0, 5.0], dtype=np.single))
    cam.lens_radius = 0.25
    o1, _ = cam.get_ray_components(8, 4, 2)
    assert not np.all(o1 == o0) and np.all(np.linalg.norm((o1 - o0).reshape(-1, 3), axis=1) <= 0.250001)
    # groups: add_sphere after a first get_hits
    grp = SphereGroup()
    grp.add_sphere(np.array([0.0, 0.0, -5.0]), 1.0, np.array([1.0, 1.0, 1.0]), 3)
    o = np.zeros((1, 3), dtype=np.single)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.single)
    assert grp.get_hits(o, d, 0.001, 1000.0)[1][0] == 4.0
    grp.add_sphere(np.array([0.0, 0.0, -2.5]), 0.5, np.array([1.0, 1.0, 1.0]), 4)
    hit = grp.get_hits(o, d, 0.001, 1000.0)
    assert hit[1][0] == 2.0 and hit[5][0] == 4


def test_stage_calls_need_a_committed_scene():
    """wrt_generate_rays before the first commit, and after a camera change without one, is a state error
    (not a kernel reading a camera that is not there)."""
    import ctypes
    import torch
    from weekend_raytracer import _native
    ctx = _native.Context(0)
    o = ctx.empty((8, 3), torch.float32)
    d = ctx.empty((8, 3), torch.float32)
    args = (ctx.handle, 4, 2, 0, 1, 7, ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctx.stream())
    assert ctx.lib.wrt_generate_rays(*args) == -3                  # WRT_E_STATE
    assert b"wrt_scene_commit" in ctx.lib.wrt_last_error(ctx.handle)
    cam = helpers.golden_scene(helpers.golden("frame_demo"))["camera"]
    ctx.upload(dict(camera=cam))
    assert ctx.lib.wrt_generate_rays(*args) == 0
    vecs = [np.ascontiguousarray(cam[k], dtype=np.double) for k in ("pos", "U", "V", "bottom_left", "horiz", "vert")]
    ptrs = [v.ctypes.data_as(ctypes.c_void_p) for v in vecs]
    assert ctx.lib.wrt_set_camera(ctx.handle, *ptrs, 0.5) == 0
    assert ctx.lib.wrt_generate_rays(*args) == -3
    assert ctx.lib.wrt_scene_commit(ctx.handle) == 0
    assert ctx.lib.wrt_generate_rays(*args) == 0
    assert ctx.lib.wrt_scene_clear(ctx.handle) == 0
    assert ctx.lib.wrt_generate_rays(*args) == -3
    # documented limits are errors, not silent truncation
    big = np.zeros(3, dtype=np.single)
    p = big.ctypes.data_as(ctypes.c_void_p)
    rc = ctx.lib.wrt_add_triangle_group(ctx.handle, p, p, p, p, p, p, p, p, p, (1 << 20) + 1, 0, p, 1.0, None)
    assert rc == -4 and b"WRT_MAX_GROUP_TRIANGLES" in ctx.lib.wrt_last_error(ctx.handle)
    rc = ctx.lib.wrt_add_sphere_group(ctx.handle, p, p, p, (1 << 24) + 1, None)
    assert rc == -4 and b"WRT_MAX_GROUP_SPHERES" in ctx.lib.wrt_last_error(ctx.handle)
    torch.cuda.synchronize()
    ctx.close()


def test_calls_leave_the_callers_device_alone():
    import torch
    from weekend_raytracer import _native
    before = torch.cuda.current_device()
    ctx = _native.Context(0)
    ctx.upload(helpers.golden_scene(helpers.golden("frame_demo")))
    accum = ctx.zeros((16, 9, 3), torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx.render(16, 9, 0, 2, 5, 1, accum)                       # a launch on a side stream ...
    again = ctx.zeros((16, 9, 3), torch.float32)
    ctx.render(16, 9, 0, 2, 5, 1, again)                           # ... and one on the current stream behind it
    ctx.upload(helpers.golden_scene(helpers.golden("frame_disks")))    # waits for both before the blob changes
    torch.cuda.synchronize()
    assert np.array_equal(accum.cpu().numpy(), again.cpu().numpy()) and accum.cpu().numpy().any()
    assert torch.cuda.current_device() == before
    ctx.close()


def test_stage_call_on_another_stream_keeps_a_running_render_ordered():
    """A render in flight on a side stream, then a stage call (get_hits) on the current stream, then a
    scene upload: the upload has to wait for the render too, not only for the stage call that
    came last, or the render would read a scene that changes under it."""
    import torch
    from weekend_raytracer import _native
    demo = helpers.golden_scene(helpers.golden("frame_demo"))
    disks = helpers.golden_scene(helpers.golden("frame_disks"))
    ctx = _native.Context(0)
    ctx.upload(demo)
    w, h, s, b, seed = 1280, 720, 24, 8, 11                    # some tens of milliseconds of rendering
    want = ctx.zeros((w, h, 3), torch.float32)
    ctx.render(w, h, 0, s, b, seed, want)
    torch.cuda.synchronize()
    rays_o = np.zeros((64, 3), dtype=np.single)
    rays_d = np.tile(np.array([[0.0, 0.0, -1.0]], dtype=np.single), (64, 1))
    for stage in ("intersect", "scatter", "generate_rays", "stats_reset"):
        ctx.upload(demo)
        got = ctx.zeros((w, h, 3), torch.float32)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ctx.render(w, h, 0, s, b, seed, got)
        if stage == "intersect":
            ctx.intersect(-1, rays_o, rays_d, 0.001, 1000.0)
        elif stage == "scatter":
            n = 64
            ctx.scatter(min(demo["materials"]), rays_d, rays_o, -rays_d, np.zeros((n, 2)), np.zeros(n, bool), np.arange(n), np.zeros(n, int),
                        np.zeros(n, int), 0, seed)
        elif stage == "generate_rays":
            ctx.generate_rays(8, 4, 0, 1, seed)
        else:
            ctx.stats_reset()
        ctx.upload(disks)                                       # must not overwrite the scene under the render
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want.cpu().numpy()), stage
    ctx.close()


def test_image_bytes_on_device_clip_and_nan():
    """wrt_image_bytes against the host quantisation (reference main.py:57-65) on values outside [0, 1],
    exact multiples of 1/255, NaN and infinities, on a ragged size."""
    from weekend_raytracer import main
    rng = np.random.default_rng(2)
    w, h = 37, 19
    image = rng.uniform(-0.2, 1.3, size=(w, h, 3)).astype(np.single)
    image[1, 2] = [np.nan, np.inf, -np.inf]
    image[3, 4] = np.arange(3, dtype=np.single) * np.single(1.0 / 255.0) * np.single(85)
    image[5, 6] = [0.0, 1.0, np.nextafter(np.single(1.0), np.single(0.0))]
    got = main.image_bytes_on_device(image)
    assert got.shape == (h, w, 3) and got.dtype == np.uint8
    assert np.array_equal(got, main.image_bytes(image))
