"""
Host logic (no GPU): the scene builders, OBJ loader, smooth normals, bounding
spheres, camera basis and texture decode of the package must hand the GPU
exactly the arrays the reference's own builders produce.  The expected arrays
are the flat scenes stored in the golden fixtures (written by
oracle/gen_golden.py from the reference's objects).
"""
import contextlib
import io

import numpy as np
import pytest

from tests import helpers
from weekend_raytracer import _flatten, main, scenes
from weekend_raytracer.obj_tri_mesh import OBJTriMesh

SCENES = {
    "demo": scenes.numpy_one_weekend_demo_scene,
    "demo_shard": scenes.numpy_one_weekend_demo_scene,
    "cover187": scenes.cover_scene,
    "glass_experiment": scenes.numpy_glass_experiment_scene,
    "simple_sphere": scenes.numpy_simple_sphere_scene,
    "triangles": scenes.numpy_triangles_scene,
    "bunny_textured": scenes.ray_group_triangle_group_bunny_scene,
    "bunny_diffuse": scenes.bunny_diffuse_scene,
    "cloth_bunny": scenes.cloth_bunny_scene,
    "texture_test": scenes.texture_test_scene,
    "smooth_normal": scenes.smooth_normal_test_scene,
    "icospheres": scenes.sphere_types_test_scene,
    "disks": scenes.disk_test_scene,
    "bunnies": scenes.numpy_bunnies_scene,
    "cow": scenes.numpy_cow_scene,
    "cow_teapot": scenes.cow_teapot_field_scene,
}


def build_flat(name, aspect):
    with contextlib.redirect_stdout(io.StringIO()):
        cam, groups, mats = SCENES[name](aspect)
    return _flatten.flatten(cam, groups, mats, main.HORIZON_COLOUR, main.SKY_COLOUR)


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(a, b), what


@pytest.mark.parametrize("name", [n for n in helpers.frame_names() if n in SCENES])
def test_scene_arrays_match_reference(name):
    f = helpers.golden("frame_" + name)
    want = helpers.golden_scene(f)
    w, h = int(f["size"][0]), int(f["size"][1])
    got = build_flat(name, w / h)
    for k in ("pos", "U", "V", "horiz", "vert", "bottom_left"):
        same(np.asarray(got["camera"][k], dtype=np.double), want["camera"][k], (name, "camera", k))
    assert float(got["camera"]["lens_radius"]) == float(want["camera"]["lens_radius"])
    assert [g["type"] for g in got["groups"]] == [g["type"] for g in want["groups"]]
    for gi, (g, e) in enumerate(zip(got["groups"], want["groups"])):
        for k, v in e.items():
            if k == "type":
                continue
            if k in ("material_id", "material_index", "radius"):
                assert g[k] == v, (name, gi, k)
            elif k == "materials":
                same(np.asarray(g[k], dtype=np.ubyte), v, (name, gi, k))
            else:
                same(np.asarray(g[k], dtype=np.single), np.asarray(v, dtype=np.single), (name, gi, k))
                if k == "bounds_radius":
                    assert np.asarray(g[k]).dtype == np.single
    assert list(got["materials"]) == list(want["materials"])
    for index, e in want["materials"].items():
        g = got["materials"][index]
        assert g["kind"] == e["kind"], (name, index)
        for k, v in e.items():
            if k == "kind":
                continue
            if k in ("fuzziness", "ior", "smallest_side"):
                assert float(g[k]) == float(v), (name, index, k)
            elif k in ("scale", "offset"):
                same(np.asarray(g[k], dtype=np.double), np.asarray(v, dtype=np.double), (name, index, k))
            else:
                same(np.asarray(g[k], dtype=np.single), np.asarray(v, dtype=np.single), (name, index, k))
    same(got["sky"]["horizon"], want["sky"]["horizon"], "horizon")
    same(got["sky"]["zenith"], want["sky"]["zenith"], "zenith")


def test_every_frame_fixture_has_its_scene_builder():
    assert sorted(SCENES) == helpers.frame_names()


@pytest.mark.parametrize("name", ["teapot", "cow"])
def test_smooth_normals_of_the_large_meshes(name):
    """The adjacency-based (linear) smooth normals against the reference's per-call scan over all faces
    (obj_tri_mesh.py:27-68) on the two large meshes, for the vertices the fixture holds."""
    f = helpers.golden("smooth_normals_" + name)
    mesh = OBJTriMesh()
    with contextlib.redirect_stdout(io.StringIO()):
        mesh.read(scenes.asset(name + ".obj"))
    assert len(f["vertices"]) >= 190
    for vertex, want in zip(f["vertices"], f["normals"]):
        got = mesh.get_smooth_vertex_normal(int(vertex))
        assert str(got.dtype) == str(f["dtype"]) and np.array_equal(got, want), int(vertex)


def test_obj_loader_conventions(tmp_path):
    path = tmp_path / "odd.obj"
    path.write_text("\n".join([
        "# comment", "v 0 0 0", "v 1 0 0", "v 0 1 0", "vt 0.5 0.25", "vn 0 0 1",
        "f 1/1/1 2//1 3",            # uv / normal absent -> -1
        "f 1 2 3 1",                 # not a triangle: skipped with a message
        "f 1 2 9",                   # index beyond what was read so far -> -1
        "v 0 0 1",
        "f 1/1 4/1 2/1",
    ]))
    mesh = OBJTriMesh()
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        mesh.read(str(path))
    assert mesh.vertices == [(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    assert mesh.uvs == [(0.5, 0.25)] and mesh.vertex_normals == [(0.0, 0.0, 1.0)]
    assert mesh.faces == [
        [(0, 0, 0), (1, -1, 0), (2, -1, -1)],
        [(0, -1, -1), (1, -1, -1), (-1, -1, -1)],
        [(0, 0, -1), (3, 0, -1), (1, 0, -1)],
    ]
    text = out.getvalue()
    assert "Non triangular face on line 8" in text and "out of range vertex (9) on line 9" in text
    assert "Read 4 vertices, 1 vertex normals, 1 UVs and 3 faces." in text
    n = mesh.get_smooth_vertex_normal(0)
    assert n.dtype == np.single and abs(float(np.linalg.norm(n)) - 1.0) < 1e-6


def test_textured_diffuse_rejects_other_modes(tmp_path):
    from PIL import Image
    from weekend_raytracer import materials
    p = tmp_path / "grey.png"
    Image.new("L", (4, 4)).save(p)
    with pytest.raises(Exception, match="Unsupported texture image mode"):
        materials.TexturedDiffuse(str(p))


def test_sphere_field_generator_is_reproducible():
    a = scenes.sphere_field_records(480)
    b = scenes.sphere_field_records(480)
    assert len(a) == 480 and a == b
    kinds = {r["material"] for r in a}
    assert kinds == {"diffuse", "metal", "glass"}
    with contextlib.redirect_stdout(io.StringIO()):
        cam, groups, mats = scenes.cover_scene(16 / 9, a)
    assert groups[0].num_spheres == 484 and max(mats) <= 254 and cam.lens_radius == 0.05


def test_image_bytes_follow_the_reference_quantisation(tmp_path, monkeypatch):
    """generate_image_from_data (reference main.py:45-74): int(255 * v) per channel, a float32 product,
    rows flipped, values outside a byte clipped as PIL does with the fill tuple, absent pixels white."""
    from PIL import Image
    rng = np.random.default_rng(3)
    w, h = 7, 5
    image = rng.uniform(-0.1, 1.2, size=(w, h, 3)).astype(np.single)
    image[0, 0] = [0.0, 1.0, np.single(1.0) / np.single(255.0) * np.single(17)]
    got = main.image_bytes(image)
    assert got.shape == (h, w, 3) and got.dtype == np.uint8
    for row in range(h):
        for column in range(w):
            want = [min(255, max(0, int(255 * v))) for v in image[column, h - 1 - row]]
            assert list(got[row, column]) == want
    monkeypatch.setattr(main, "IMG_WIDTH", w)
    monkeypatch.setattr(main, "IMG_HEIGHT", h)
    pixels = {(x, y): image[x, y] for x in range(w) for y in range(h) if (x, y) != (2, 3)}
    main.generate_image_from_data(pixels, path=str(tmp_path / "out.png"))
    stored = np.asarray(Image.open(tmp_path / "out.png"))
    want = got.copy()
    want[h - 1 - 3, 2] = 255
    assert np.array_equal(stored, want)
