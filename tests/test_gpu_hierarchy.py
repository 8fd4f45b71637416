"""
Walking a triangle group through its box hierarchy (wrt_set_hierarchy) must find the
closest hit the brute-force loops find, bit for bit, ties included: whole frames
both ways are compared path by path, single rays (axis-parallel ones, rays that start
on a surface, rays along edges and through corners among them) hit by hit, on the
shipped meshes, on scaled and shifted meshes against the oracle, and at the benchmark's
full size.
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import numpy_path
from tests.test_gpu_reject import Flat, mesh_scene

pytestmark = pytest.mark.gpu


def scene_objects(fn, aspect):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(aspect)


def both_ways(objs, w, h, s, b, seed):
    import torch
    from weekend_raytracer import main
    out = []
    for hierarchy in (True, False):
        frame = main.Frame(objs, hierarchy=16 if hierarchy else 0)
        frame.ctx.stats_reset()
        paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
        frame.accumulate(w, h, 0, s, b, seed, paths=paths)
        out.append((paths.cpu().numpy(), frame.ctx.stats()))
        frame.ctx.close()
    return out


@pytest.mark.parametrize("name,size", [
    ("ray_group_triangle_group_bunny_scene", (320, 180, 6, 10)),
    ("cloth_bunny_scene", (256, 144, 4, 8)),
    ("cow_teapot_field_scene", (320, 180, 4, 8)),
    ("bunny_diffuse_scene", (1920, 1080, 30, 10)),        # the benchmark's mesh configuration, full size
])
def test_hierarchy_changes_nothing(name, size):
    from weekend_raytracer import scenes
    w, h, s, b = size
    objs = scene_objects(getattr(scenes, name), w / h)
    (p_on, st_on), (p_off, st_off) = both_ways(objs, w, h, s, b, seed=2025)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"])
    # the pairs the reference would have examined are counted the same; far fewer are evaluated
    assert st_on["triangle_pairs"] == st_off["triangle_pairs"] > 0
    assert 0 < st_on["triangle_evaluations"] < 0.1 * st_on["triangle_pairs"]


@pytest.mark.parametrize("scale,offset", [
    (1.0, (0.0, 0.0, 0.0)),
    (0.01, (0.0, 0.0, 0.0)),
    (1.0, (300.0, -200.0, 150.0)),
    (0.06, (8000.0, -3000.0, 500.0)),
    (30.0, (0.0, 0.0, 0.0)),                   # edges of ~10 units: the boxes grow by more than the mesh's size
])
def test_hierarchy_on_scaled_meshes_matches_oracle(scale, offset):
    from weekend_raytracer import main
    scene = mesh_scene(scale, np.array(offset))
    w, h, s, b = 60, 40, 3, 8
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    main_hierarchy = main.HIERARCHY
    main.HIERARCHY = 16
    try:
        got = main.render_frame(objs, w, h, s, b, seed=11, want_paths=True)
    finally:
        main.HIERARCHY = main_hierarchy
    want = numpy_path.render(scene, w, h, s, b, seed=11)
    assert np.array_equal(got["path_colours"], want["path_colours"])
    assert np.array_equal(got["counts"], want["counts"])


def awkward_rays(pt0, A, B, rng, n_random):
    """Rays that stress the box tests: axis-parallel directions (zero components, both signs of zero),
    origins on the surfaces, rays through corners and along edges, grazing rays in triangle planes."""
    lo, hi = pt0.min(axis=0) - 1.0, pt0.max(axis=0) + 1.0
    o = [rng.uniform(lo, hi, (n_random, 3))]
    d = [rng.normal(size=(n_random, 3))]
    pick = rng.integers(0, len(pt0), 400)
    corner = pt0[pick]
    inside = pt0[pick] + 0.3 * A[pick] + 0.3 * B[pick]
    start = rng.uniform(lo, hi, (400, 3))
    o += [start, start]                                   # towards corners and interior points
    d += [corner - start, inside - start]
    o += [inside, corner]                                 # starting on a surface / a corner
    d += [rng.normal(size=(400, 3)), rng.normal(size=(400, 3))]
    o += [corner - 2.0 * A[pick], inside - 1.5 * (A[pick] + 0.2 * B[pick])]   # along an edge, in the plane
    d += [A[pick], A[pick] + 0.2 * B[pick]]
    for axis in range(3):
        for sign in (1.0, -1.0):
            for zero in (0.0, -0.0):
                dirs = np.full((300, 3), zero)
                dirs[:, axis] = sign
                origin = inside[:300].copy()
                origin[:, axis] -= sign * 3.0
                o.append(origin)
                d.append(dirs)
                o.append(rng.uniform(lo, hi, (300, 3)))
                d.append(dirs)
    o, d = np.concatenate(o), np.concatenate(d)
    d[:n_random] /= np.linalg.norm(d[:n_random], axis=1)[:, None]
    return o.astype(np.single), d.astype(np.single)


@pytest.mark.parametrize("name", ["ray_group_triangle_group_bunny_scene", "cow_teapot_field_scene"])
def test_single_rays_hit_the_same_triangle(name):
    from weekend_raytracer import main, scenes
    objs = scene_objects(getattr(scenes, name), 16 / 9)
    rng = np.random.default_rng(5)
    groups = [g for g in objs[1] if hasattr(g, "pt0s") and len(g.pt0s) >= 64]
    assert groups
    g = groups[0]
    o, d = awkward_rays(np.asarray(g.pt0s), np.asarray(g.As), np.asarray(g.Bs), rng, 20000)
    res = []
    for hierarchy in (True, False):
        frame = main.Frame(objs, hierarchy=16 if hierarchy else 0)
        frame.ctx.stats_reset()
        res.append(frame.ctx.intersect(-1, o, d, np.single(0.001), np.single(1000.0)))
        frame.ctx.close()
    on, off = res
    assert off["hit"].mean() > 0.2
    for key in ("hit", "group", "prim", "back", "mat"):
        assert np.array_equal(on[key], off[key]), key
    for key in ("t", "p", "n", "uv"):
        assert np.array_equal(on[key].view(np.uint32), off[key].view(np.uint32)), key


def test_rays_outside_the_walks_range_fall_back():
    """Origins beyond 1e9, directions shorter than 1e-9 or longer than 1e9 and NaN rays are outside the
    range the box growth is proven for: their wavefronts go through the loops, with the same answers."""
    from weekend_raytracer import main
    scene = mesh_scene(1.0, np.zeros(3))
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    rng = np.random.default_rng(9)
    n = 4096
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    far = rng.integers(0, n, 40)
    o[far] = -d[far] * 3.0e9                       # aimed at the mesh from 3e9 away
    d[rng.integers(0, n, 40)] *= 1.0e-10
    d[rng.integers(0, n, 40)] *= 1.0e10
    d[rng.integers(0, n, 20), 1] = np.nan
    o[rng.integers(0, n, 20), 0] = np.inf
    o, d = o.astype(np.single), d.astype(np.single)
    res = []
    for hierarchy in (16, 0):
        frame = main.Frame(objs, hierarchy=hierarchy)
        res.append(frame.ctx.intersect(-1, o, d, np.single(0.001), np.single(1000.0)))
        frame.ctx.close()
    on, off = res
    assert 0.05 < off["hit"].mean() < 0.95
    for key in ("hit", "group", "prim", "back", "mat"):
        assert np.array_equal(on[key], off[key]), key
    for key in ("t", "p", "n", "uv"):
        assert np.array_equal(on[key].view(np.uint32), off[key].view(np.uint32)), key


def test_meshes_the_builder_declines():
    """A group with a corner that is not finite gets no hierarchy (the loops handle it as before); a group
    with triangles of zero area gets one."""
    from weekend_raytracer import main
    for spoil in ("nan", "flat"):
        scene = mesh_scene(1.0, np.zeros(3))
        tris = scene["groups"][1]
        if spoil == "nan":
            tris["A"] = tris["A"].copy()
            tris["A"][7, 1] = np.nan
        else:
            tris["B"] = tris["B"].copy()
            tris["B"][5] = tris["A"][5] * 2.0          # B parallel to A: zero area
            tris["A"][9] = 0.0                         # a point
        objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
                {i: Flat(m) for i, m in scene["materials"].items()})
        (p_on, st_on), (p_off, st_off) = both_ways(objs, 60, 40, 2, 6, seed=3)
        assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
        assert st_on["triangle_pairs"] == st_off["triangle_pairs"]


@pytest.mark.parametrize("kind", ["triangles", "spheres"])
def test_groups_with_leaves_of_eight(kind):
    """Above 16 384 primitives the builder puts up to eight in a leaf, and the walk gives a pooled leaf eight
    lanes instead of four: 20 000 overlapping triangles, 20 000 small spheres (with the ground and two huge
    ones kept out of the tree), frames with the hierarchy against frames from the loops."""
    from tests.test_gpu_reject import sphere_scene
    if kind == "triangles":
        scene = mesh_scene(1.0, np.zeros(3), n_tri=20000, seed=11)
        w, h, s, b = 96, 54, 2, 5
    else:
        scene = sphere_scene(1.0, np.zeros(3), n=20000, seed=9)
        scene["groups"][0]["radii"][3:] *= np.single(0.08)           # a cloud of small spheres instead of a solid lump
        w, h, s, b = 160, 90, 2, 6
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    (p_on, st_on), (p_off, st_off) = both_ways(objs, w, h, s, b, seed=5)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"])
    if kind == "triangles":
        # (triangles with edges a third of the whole mesh's radius: their boxes overlap a lot)
        assert st_on["triangle_pairs"] == st_off["triangle_pairs"] > 0
        assert 0 < st_on["triangle_evaluations"] < 0.5 * st_on["triangle_pairs"]
    else:
        assert st_on["sphere_pairs"] == st_off["sphere_pairs"] > 0
        assert 0 < st_on["sphere_evaluations"] < 0.5 * st_on["sphere_pairs"]


@pytest.mark.parametrize("n_copies,n_per_group", [(6, 40), (40, 24)])
def test_ties_between_walked_groups(n_copies, n_per_group):
    """The same spheres and triangles in many groups, every one of them walked: each hit is a tie between
    groups, which the earliest group wins whatever order the walks run in; with more than 32 large groups
    the further ones are walked in scene order, out of the per-lane pass over the groups."""
    from tests.test_gpu_random_scenes import layered_scene
    from weekend_raytracer import main
    scene = layered_scene(n_copies, n_per_group, seed=n_copies)
    w, h, s, b = 48, 30, 2, 6
    want = numpy_path.render(scene, w, h, s, b, seed=78)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    frame = main.Frame(objs, hierarchy=16)
    frame.ctx.stats_reset()
    import torch
    paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
    frame.accumulate(w, h, 0, s, b, 78, paths=paths)
    stats = frame.ctx.stats()
    frame.ctx.close()
    got = paths.cpu().numpy()
    assert np.array_equal(got[:, :3], want["path_colours"])
    from weekend_raytracer import _native
    assert np.array_equal(got[:, 3].astype(np.int32) % _native.SURVIVOR_MARK, want["segments"])
    # walked, not looped: fewer primitives evaluated than pairs examined (two dozen large triangles in a
    # group leave little to skip)
    assert 0 < stats["triangle_evaluations"] < 0.8 * stats["triangle_pairs"]
    assert 0 < stats["sphere_evaluations"] < 0.8 * stats["sphere_pairs"]


# ---- sphere groups ---------------------------------------------------------------------------
@pytest.mark.parametrize("field,size", [
    (None, (320, 180, 6, 10)),                # the shipped 187-sphere field (+ ground and three unit spheres)
    (480, (1280, 720, 10, 10)),               # the benchmark's field at its full frame size, 10 of its 100 spp
])
def test_sphere_hierarchy_changes_nothing(field, size):
    """A sphere group walked through its hierarchy (the ground sphere kept out of the tree) gives the frame
    of the uniform loop bit for bit: depth-of-field camera, diffuse / fuzzy metal / glass spheres."""
    from weekend_raytracer import scenes
    w, h, s, b = size
    with contextlib.redirect_stdout(io.StringIO()):
        objs = scenes.cover_scene(w / h, None if field is None else scenes.sphere_field_records(field))
    (p_on, st_on), (p_off, st_off) = both_ways(objs, w, h, s, b, seed=77)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"])
    assert st_on["sphere_pairs"] == st_off["sphere_pairs"] > 0


@pytest.mark.parametrize("scale,offset", [
    (1.0, (0.0, 0.0, 0.0)),
    (0.01, (0.0, 0.0, 0.0)),
    (100.0, (0.0, 0.0, 0.0)),
    (1.0, (300.0, -200.0, 150.0)),
    (0.05, (8000.0, -3000.0, 500.0)),
])
def test_sphere_hierarchy_on_scaled_scenes_matches_oracle(scale, offset):
    """Small spheres, two huge ones (kept out of the tree), spheres inside spheres, far from the origin."""
    from tests.test_gpu_reject import sphere_scene
    from weekend_raytracer import main
    scene = sphere_scene(scale, np.array(offset))
    w, h, s, b = 60, 40, 3, 8
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    main_hierarchy = main.HIERARCHY
    main.HIERARCHY = 16
    try:
        got = main.render_frame(objs, w, h, s, b, seed=12, want_paths=True)
    finally:
        main.HIERARCHY = main_hierarchy
    want = numpy_path.render(scene, w, h, s, b, seed=12)
    assert np.array_equal(got["path_colours"], want["path_colours"])
    assert np.array_equal(got["counts"], want["counts"])


def test_single_rays_hit_the_same_sphere():
    """Rays grazing spheres, starting on and inside them, axis-parallel, unnormalised (within and beyond the
    range the growth is proven for: the latter fall back), against the uniform loop, record by record."""
    from tests.test_gpu_reject import sphere_scene
    from weekend_raytracer import main
    scene = sphere_scene(1.0, np.zeros(3), n=300, seed=8)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    c = scene["groups"][0]["centres"].astype(np.double)
    r = scene["groups"][0]["radii"].astype(np.double)
    rng = np.random.default_rng(6)
    n = 20000
    o = rng.uniform(-5, 5, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pick = rng.integers(3, len(r), n)
    tangent = np.cross(d, rng.normal(size=(n, 3)))
    tangent /= np.linalg.norm(tangent, axis=1)[:, None]
    kind = rng.integers(0, 6, n)
    graze = c[pick] + tangent * (r[pick] * rng.uniform(0.999, 1.001, n))[:, None]    # lines that skim a sphere
    o = np.where((kind == 1)[:, None], graze - 4.0 * d, o)
    on_surface = c[pick] + tangent * r[pick][:, None]
    o = np.where((kind == 2)[:, None], on_surface, o)
    o = np.where((kind == 3)[:, None], c[pick] + 0.3 * tangent * r[pick][:, None], o)  # inside
    axis = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], n)[:, None]
    d = np.where((kind == 4)[:, None], axis, d)
    d = np.where((kind == 5)[:, None], d * rng.uniform(0.9990, 1.0010, n)[:, None], d)   # not unit length
    d[:50] *= 3.0                                                                         # far from unit length
    o, d = o.astype(np.single), d.astype(np.single)
    res = []
    for hierarchy in (16, 0):
        frame = main.Frame(objs, hierarchy=hierarchy)
        res.append(frame.ctx.intersect(-1, o, d, np.single(0.001), np.single(1000.0)))
        frame.ctx.close()
    on, off = res
    assert off["hit"].mean() > 0.3
    for key in ("hit", "group", "prim", "back", "mat"):
        assert np.array_equal(on[key], off[key]), key
    for key in ("t", "p", "n", "uv"):
        assert np.array_equal(on[key].view(np.uint32), off[key].view(np.uint32)), key
