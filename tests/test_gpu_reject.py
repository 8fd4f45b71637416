"""
The bounding-sphere reject in front of the triangle evaluation must never change
a result: whole frames with it switched on and off are compared path by path,
bit for bit, on meshes, on scaled copies of a mesh (the error bounds of the test
scale with the scene), on a group whose triangles are too large for the bounds
(the wavefronts then fall back to the full evaluation) and at the benchmark's
full size.
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import numpy_path
from tests import helpers

pytestmark = pytest.mark.gpu


def both_ways(objs, w, h, s, b, seed):
    import torch
    from weekend_raytracer import main
    out = []
    for reject in (True, False):
        frame = main.Frame(objs, hierarchy=0)        # the loops themselves, whatever WRT_HIERARCHY presets
        frame.ctx.set_triangle_reject(reject)
        frame.ctx.stats_reset()
        paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
        frame.accumulate(w, h, 0, s, b, seed, paths=paths)
        stats = frame.ctx.stats()
        out.append((paths.cpu().numpy(), stats))
        frame.ctx.set_triangle_reject(True)
        frame.ctx.close()
    return out


def scene_objects(fn, aspect):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(aspect)


@pytest.mark.parametrize("name,size", [
    ("ray_group_triangle_group_bunny_scene", (320, 180, 6, 10)),
    ("cloth_bunny_scene", (256, 144, 4, 8)),
    ("cow_teapot_field_scene", (160, 90, 2, 6)),
    ("bunny_diffuse_scene", (1920, 1080, 30, 10)),        # the benchmark's mesh configuration, full size
])
def test_reject_changes_nothing(name, size):
    from weekend_raytracer import scenes
    w, h, s, b = size
    objs = scene_objects(getattr(scenes, name), w / h)
    (p_on, st_on), (p_off, st_off) = both_ways(objs, w, h, s, b, seed=2024)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"])
    assert st_on["triangle_pairs"] == st_off["triangle_pairs"] > 0
    # the reject leaves a small share of the pairs to the full evaluation; off, none is left to it
    assert 0 < st_on["triangle_evaluations"] < 0.2 * st_on["triangle_pairs"]
    assert st_off["triangle_evaluations"] == 0


def mesh_scene(scale, offset, n_tri=200, seed=3):
    """A closed-ish random fan mesh around `offset`, edge lengths ~0.3 * scale, glass and diffuse."""
    rng = np.random.default_rng(seed)
    theta = rng.uniform(0, 2 * np.pi, n_tri)
    phi = np.arccos(rng.uniform(-1, 1, n_tri))
    centre = np.stack([np.sin(phi) * np.cos(theta), np.cos(phi), np.sin(phi) * np.sin(theta)], axis=1)
    p0 = ((centre + rng.normal(scale=0.05, size=(n_tri, 3))) * scale + offset).astype(np.single)
    A = (rng.normal(scale=0.3, size=(n_tri, 3)) * scale).astype(np.single)
    B = (rng.normal(scale=0.3, size=(n_tri, 3)) * scale).astype(np.single)
    nrm = np.cross(A, B)
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.single)
    uv = np.zeros((n_tri, 2), np.single)
    pts = np.concatenate([p0, p0 + A, p0 + B])
    bc = np.mean(pts, axis=0).astype(np.single)
    br = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
    tris = dict(type="triangles", pt0=p0, A=A, B=B, n0=nrm, n1=nrm, n2=nrm, uv0=uv, uv1=uv, uv2=uv, material_id=1,
                bounds_centre=bc, bounds_radius=br)
    ground = dict(type="spheres", centres=(np.array([[0, -1001.0, 0]]) * scale + offset).astype(np.single),
                  radii=np.array([1000.0 * scale], np.single), materials=np.array([0], np.ubyte))
    materials = {0: dict(kind="diffuse", colour=np.array([0.6, 0.6, 0.6], np.single)), 1: dict(kind="dielectric", ior=1.5)}
    cam = numpy_path.camera_basis(np.array([0.5, 1.0, 4.0]) * scale + offset, offset + np.array([0, 0.0, 0.0]),
                                  4.0 * scale, 0.05 * scale, 1.5, 50.0)
    sky = dict(horizon=np.array([1, 1, 1], np.single), zenith=np.array([0.5, 0.7, 1.0], np.single))
    return dict(camera=cam, groups=[ground, tris], materials=materials, sky=sky)


class Flat:
    def __init__(self, rec):
        self.rec = rec

    def _flat(self):
        return self.rec


@pytest.mark.parametrize("scale,offset,falls_back", [
    (1.0, (0.0, 0.0, 0.0), False),
    (0.01, (0.0, 0.0, 0.0), False),
    (1.0, (300.0, -200.0, 150.0), False),      # far from the origin: large coordinates, small triangles
    (0.06, (8000.0, -3000.0, 500.0), False),   # very far: the reject must neither miss a hit nor give up rejecting
    (30.0, (0.0, 0.0, 0.0), True),             # edges of ~10 units: outside the reject's error bounds
])
def test_reject_on_scaled_meshes_matches_oracle(scale, offset, falls_back):
    scene = mesh_scene(scale, np.array(offset))
    w, h, s, b = 60, 40, 3, 8
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    (p_on, st_on), (p_off, st_off) = both_ways(objs, w, h, s, b, seed=11)
    assert np.array_equal(p_on.view(np.uint32), p_off.view(np.uint32))
    assert (st_on["triangle_evaluations"] == 0) == falls_back
    assert st_on["triangle_evaluations"] < 0.3 * st_on["triangle_pairs"]
    want = numpy_path.render(scene, w, h, s, b, seed=11)
    same = np.all(p_on[:, :3] == want["path_colours"], axis=1)
    assert same.all(), same.mean()


def test_reruns_are_bit_identical():
    """Which wavefront traces a batch depends on timing (batches come from one counter); every
    path's record must not."""
    import torch
    from weekend_raytracer import main, scenes
    w, h, s, b = 640, 360, 8, 10
    objs = scene_objects(scenes.cloth_bunny_scene, w / h)
    frame = main.Frame(objs)
    runs = []
    for _ in range(3):
        paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
        accum = frame.accumulate(w, h, 0, s, b, 99, paths=paths)
        runs.append((paths.cpu().numpy().view(np.uint32), accum.cpu().numpy().view(np.uint32)))
    frame.ctx.close()
    for p, a in runs[1:]:
        assert np.array_equal(p, runs[0][0]) and np.array_equal(a, runs[0][1])


def sphere_scene(scale, offset, n=60, seed=4):
    """One large sphere group holding small spheres, a few huge ones and the ground, about `offset`."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-3, 3, (n, 3)) * np.array([1, 0.3, 1]) + np.array([0, 0.6, 0])
    r = rng.uniform(0.1, 0.5, n)
    c[0], r[0] = (0.0, -1000.0, 0.0), 1000.0                 # ground, inside the same group
    c[1], r[1] = (0.0, 0.0, -30.0), 25.0                     # a big one behind
    c[2], r[2] = (0.2, 0.9, 0.3), 0.9                        # one the others are partly inside of
    centres = (c * scale + offset).astype(np.single)
    radii = (r * scale).astype(np.single)
    mats = (np.arange(n) % 3).astype(np.ubyte)
    materials = {0: dict(kind="diffuse", colour=np.array([0.6, 0.5, 0.4], np.single)),
                 1: dict(kind="dielectric", ior=1.5),
                 2: dict(kind="metal", colour=np.array([0.9, 0.9, 0.9], np.single), fuzziness=0.2)}
    cam = numpy_path.camera_basis(np.array([4.0, 2.0, 7.0]) * scale + offset, np.array([0.0, 0.6, 0.0]) * scale + offset,
                                  8.0 * scale, 0.1 * scale, 1.5, 45.0)
    sky = dict(horizon=np.array([1, 1, 1], np.single), zenith=np.array([0.5, 0.7, 1.0], np.single))
    return dict(camera=cam, groups=[dict(type="spheres", centres=centres, radii=radii, materials=mats)],
                materials=materials, sky=sky)


@pytest.mark.parametrize("scale,offset", [
    (1.0, (0.0, 0.0, 0.0)),
    (0.01, (0.0, 0.0, 0.0)),
    (100.0, (0.0, 0.0, 0.0)),
    (1.0, (300.0, -200.0, 150.0)),
    (0.05, (8000.0, -3000.0, 500.0)),
])
def test_sphere_bound_on_scaled_scenes_matches_oracle(scale, offset):
    """The sphere loops vote with an upper bound of the discriminant computed about the group's
    median centre; it must never miss a sphere the reference's own arithmetic hits."""
    from weekend_raytracer import main
    scene = sphere_scene(scale, np.array(offset))
    w, h, s, b = 64, 40, 3, 8
    want = numpy_path.render(scene, w, h, s, b, seed=21)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    got = main.render_frame(objs, w, h, s, b, seed=21, want_paths=True)
    assert np.array_equal(got["path_colours"], want["path_colours"])
    assert np.array_equal(got["counts"], want["counts"])
    assert want["counts"][1] > 0.5 * want["counts"][0]
