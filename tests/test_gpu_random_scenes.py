"""
Randomly generated scenes through the C ABI against the CPU oracle: group
sizes around every code-path boundary of the render kernel (per-lane groups of
up to 8 primitives, odd and even loop tails, many large groups so that lanes
park at different groups), all material kinds, disks, depth of field, ragged
image sizes and sample offsets.
"""
import numpy as np
import pytest

from oracle import numpy_path
from tests import helpers

pytestmark = pytest.mark.gpu


def random_scene(rng, n_groups, sizes):
    kinds = ["diffuse", "checker", "normal_rgb", "normal_discrete", "metal", "metal_fuzzy", "dielectric", "textured"]
    materials = {}
    tex = helpers.load_texture("uv_test.jpg")
    for i, kind in enumerate(kinds):
        if kind == "diffuse":
            materials[i] = dict(kind="diffuse", colour=rng.uniform(0.2, 0.9, 3).astype(np.single))
        elif kind == "checker":
            materials[i] = dict(kind="checker", scale=rng.uniform(1, 4, 3), offset=rng.uniform(0, 1, 3),
                                colour_a=rng.uniform(0.5, 1, 3), colour_b=rng.uniform(0, 0.5, 3))
        elif kind == "metal":
            materials[i] = dict(kind="metal", colour=rng.uniform(0.6, 1, 3).astype(np.single), fuzziness=0.0)
        elif kind == "metal_fuzzy":
            materials[i] = dict(kind="metal", colour=rng.uniform(0.6, 1, 3).astype(np.single),
                                fuzziness=float(rng.uniform(0.05, 0.6)))
        elif kind == "dielectric":
            materials[i] = dict(kind="dielectric", ior=float(rng.uniform(1.2, 1.8)))
        elif kind == "textured":
            materials[i] = dict(kind="textured", pixels=tex, smallest_side=float(min(tex.shape[:2])))
        else:
            materials[i] = dict(kind=kind)
    groups = [dict(type="spheres", centres=np.array([[0, -500.5, 0]], np.single), radii=np.array([500.0], np.single),
                   materials=np.array([1], np.ubyte))]
    for gi in range(n_groups):
        n = sizes[gi % len(sizes)]
        centre = rng.uniform(-3, 3, 3) * np.array([1, 0.3, 1]) + np.array([0, 1.0, 0])
        which = rng.integers(0, 3)
        if which == 0:
            c = (centre + rng.uniform(-1, 1, (n, 3))).astype(np.single)
            groups.append(dict(type="spheres", centres=c, radii=rng.uniform(0.1, 0.5, n).astype(np.single),
                               materials=rng.integers(0, len(kinds) + 1, n).astype(np.ubyte)))   # one index unmapped
        elif which == 1:
            p0 = (centre + rng.uniform(-1, 1, (n, 3))).astype(np.single)
            A = rng.uniform(-0.7, 0.7, (n, 3)).astype(np.single)
            B = rng.uniform(-0.7, 0.7, (n, 3)).astype(np.single)

            def unit(v):
                return (v / np.linalg.norm(v, axis=1)[:, None]).astype(np.single)
            nrm = [unit(np.cross(A, B) + rng.normal(scale=0.2, size=(n, 3))) for _ in range(3)]
            uv = [rng.uniform(0, 1, (n, 2)).astype(np.single) for _ in range(3)]
            pts = np.concatenate([p0, p0 + A, p0 + B])
            bc = np.mean(pts, axis=0).astype(np.single)
            br = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
            groups.append(dict(type="triangles", pt0=p0, A=A, B=B, n0=nrm[0], n1=nrm[1], n2=nrm[2], uv0=uv[0],
                               uv1=uv[1], uv2=uv[2], material_id=int(rng.integers(0, len(kinds))), bounds_centre=bc,
                               bounds_radius=br))
        else:
            nv = rng.normal(size=3)
            nv = (nv / np.linalg.norm(nv)).astype(np.single)
            up = np.array([0, 1, 0], np.single) if abs(nv[1]) < 0.9 else np.array([1, 0, 0], np.single)
            U = np.cross(up, nv)
            U = U / np.linalg.norm(U)
            V = np.cross(nv, U)
            groups.append(dict(type="disk", centre=centre.astype(np.single), normal=nv, U=U.astype(np.single),
                               V=V.astype(np.single), radius=float(rng.uniform(0.3, 1.2)),
                               material_index=int(rng.integers(0, len(kinds)))))
    cam = numpy_path.camera_basis(np.array([6.0, 3.0, 7.0]), np.array([0.0, 0.8, 0.0]), 9.0,
                                  float(rng.choice([0.0, 0.3])), 1.7, 45.0)
    sky = dict(horizon=np.array([1, 1, 1], np.single), zenith=np.array([0.5, 0.7, 1.0], np.single))
    return dict(camera=cam, groups=groups, materials=materials, sky=sky)


class Flat:
    def __init__(self, rec):
        self.rec = rec

    def _flat(self):
        return self.rec


@pytest.mark.parametrize("seed,n_groups,sizes", [
    (1, 3, [1, 2, 3]),
    (2, 4, [7, 8, 9, 10]),            # around the per-lane / parked boundary
    (3, 5, [11, 12, 13, 30, 31]),     # odd / even tails of the two-per-trip loops
    (4, 9, [20, 9, 33]),              # lanes park at many different groups
    (5, 12, [4, 40, 1, 17]),
])
def test_random_scene_matches_oracle(seed, n_groups, sizes):
    from weekend_raytracer import main
    rng = np.random.default_rng(seed)
    scene = random_scene(rng, n_groups, sizes)
    w, h, s, b, s_begin = 45 + seed, 27, 3, 7, seed
    want = numpy_path.render(scene, w, h, s, b, seed=1000 + seed, s_begin=s_begin)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    got = main.render_frame(objs, w, h, s, b, seed=1000 + seed, s_begin=s_begin, want_paths=True)
    same = np.all(got["path_colours"] == want["path_colours"], axis=1)
    assert same.all(), same.mean()
    assert np.array_equal(got["segments"], want["segments"])
    assert helpers.psnr(np.nan_to_num(got["image"]), np.nan_to_num(want["image"])) >= 40.0
    if same.all():
        assert np.array_equal(got["counts"], want["counts"])
    # roofline bookkeeping equals what the reference would evaluate
    segments = int(got["counts"][:-1].sum())
    n_sph = sum(g["radii"].shape[0] for g in scene["groups"] if g["type"] == "spheres")
    n_tri_groups = sum(1 for g in scene["groups"] if g["type"] == "triangles")
    n_disks = sum(1 for g in scene["groups"] if g["type"] == "disk")
    st = got["stats"]
    assert st["sphere_pairs"] == segments * n_sph
    assert st["cull_tests"] == segments * n_tri_groups
    assert st["disk_pairs"] == segments * n_disks


def test_intersect_entry_point_on_random_scene():
    from weekend_raytracer import _native
    rng = np.random.default_rng(77)
    scene = random_scene(rng, 8, [9, 2, 25, 16])
    ctx = _native.Context(0)
    ctx.upload(scene)
    o = rng.uniform(-6, 6, (5000, 3)).astype(np.single)
    d = rng.normal(size=(5000, 3))
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.single)
    got = ctx.intersect(-1, o, d, 0.001, 1000.0)
    want = numpy_path.closest_hit(scene["groups"], o, d)
    assert np.array_equal(got["hit"], want["hit"])
    hit = want["hit"]
    for k in ("t", "p", "n", "uv"):
        assert np.array_equal(got[k][hit], want[k][hit].astype(np.single)), k
    for k in ("mat", "back", "group", "prim"):
        assert np.array_equal(got[k][hit], want[k][hit]), k
    ctx.close()


def stacked_scene(n_spheres, n_triangles):
    """Concentric spheres and a deck of parallel triangles: every ray through the middle is a
    candidate for dozens of primitives of one group, far more than a lane's queue holds."""
    materials = {0: dict(kind="diffuse", colour=np.array([0.7, 0.6, 0.5], np.single)),
                 1: dict(kind="dielectric", ior=1.5), 2: dict(kind="metal", colour=np.array([0.9, 0.9, 0.9], np.single),
                                                             fuzziness=0.0)}
    radii = np.linspace(0.2, 1.4, n_spheres).astype(np.single)[::-1].copy()      # outermost first
    spheres = dict(type="spheres", centres=np.tile(np.array([[-1.5, 1.0, 0.0]], np.single), (n_spheres, 1)),
                   radii=radii, materials=(np.arange(n_spheres) % 3).astype(np.ubyte))
    z = np.linspace(-1.0, 1.0, n_triangles).astype(np.single)
    p0 = np.stack([np.full(n_triangles, 0.5, np.single), np.full(n_triangles, 0.0, np.single), z], axis=1)
    A = np.tile(np.array([[2.0, 0.0, 0.0]], np.single), (n_triangles, 1))
    B = np.tile(np.array([[1.0, 2.0, 0.0]], np.single), (n_triangles, 1))
    nrm = np.tile(np.array([[0.0, 0.0, 1.0]], np.single), (n_triangles, 1))
    uv = np.zeros((n_triangles, 2), np.single)
    pts = np.concatenate([p0, p0 + A, p0 + B])
    bc = np.mean(pts, axis=0).astype(np.single)
    br = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
    tris = dict(type="triangles", pt0=p0, A=A, B=B, n0=nrm, n1=nrm, n2=nrm, uv0=uv, uv1=uv, uv2=uv, material_id=1,
                bounds_centre=bc, bounds_radius=br)
    ground = dict(type="spheres", centres=np.array([[0, -500.0, 0]], np.single), radii=np.array([500.0], np.single),
                  materials=np.array([0], np.ubyte))
    cam = numpy_path.camera_basis(np.array([0.3, 1.2, 6.0]), np.array([0.0, 1.0, 0.0]), 6.0, 0.0, 1.6, 40.0)
    sky = dict(horizon=np.array([1, 1, 1], np.single), zenith=np.array([0.5, 0.7, 1.0], np.single))
    return dict(camera=cam, groups=[ground, spheres, tris], materials=materials, sky=sky)


@pytest.mark.parametrize("n_spheres,n_triangles", [(37, 41), (64, 100), (9, 9)])
def test_more_candidates_than_queue_slots(n_spheres, n_triangles):
    from weekend_raytracer import _native, main
    scene = stacked_scene(n_spheres, n_triangles)
    rng = np.random.default_rng(n_spheres)
    # rays along +-z through the deck and through the spheres' common centre, plus random ones
    n = 4096
    o = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(0.1, 2.0, n), np.where(rng.random(n) < 0.5, -5.0, 5.0)],
                 axis=1).astype(np.single)
    target = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(0.1, 2.0, n), np.zeros(n)], axis=1)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.single)
    ctx = _native.Context(0)
    ctx.upload(scene)
    got = ctx.intersect(-1, o, d, 0.001, 1000.0)
    ctx.close()
    want = numpy_path.closest_hit(scene["groups"], o, d)
    assert np.array_equal(got["hit"], want["hit"])
    hit = want["hit"]
    assert hit.mean() > 0.5
    assert np.array_equal(got["t"][hit], want["t"][hit].astype(np.single))
    for k in ("group", "prim", "mat", "back"):
        assert np.array_equal(got[k][hit], want[k][hit]), k
    # and through the bounce kernel (glass deck: many refractions in a row)
    w, h, s, b = 40, 25, 2, 12
    ref = numpy_path.render(scene, w, h, s, b, seed=5)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    out = main.render_frame(objs, w, h, s, b, seed=5, want_paths=True)
    same = np.all(out["path_colours"] == ref["path_colours"], axis=1)
    assert same.all(), same.mean()
    assert np.array_equal(out["segments"], ref["segments"])


def layered_scene(n_copies, n_per_group, seed):
    """The same primitives in several large groups with different materials: every hit is a tie
    between groups, which the earliest group must win whatever order the groups are run in; with
    more than 32 large groups some of them are gone through in scene order."""
    rng = np.random.default_rng(seed)
    materials = {0: dict(kind="diffuse", colour=np.array([0.5, 0.5, 0.5], np.single))}
    groups = [dict(type="spheres", centres=np.array([[0, -500.5, 0]], np.single), radii=np.array([500.0], np.single),
                   materials=np.array([0], np.ubyte))]
    c = rng.uniform(-2, 2, (n_per_group, 3)).astype(np.single) * np.array([1, 0.4, 1], np.single) + np.array([0, 1, 0], np.single)
    rad = rng.uniform(0.15, 0.4, n_per_group).astype(np.single)
    p0 = (rng.uniform(-2, 2, (n_per_group, 3)) * np.array([1, 0.4, 1]) + np.array([0, 1.2, 0])).astype(np.single)
    A = rng.uniform(-0.8, 0.8, (n_per_group, 3)).astype(np.single)
    B = rng.uniform(-0.8, 0.8, (n_per_group, 3)).astype(np.single)
    nrm = np.cross(A, B)
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.single)
    uv = np.zeros((n_per_group, 2), np.single)
    pts = np.concatenate([p0, p0 + A, p0 + B])
    bc = np.mean(pts, axis=0).astype(np.single)
    br = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
    for k in range(n_copies):
        m = 1 + k
        kind = k % 3
        materials[m] = (dict(kind="diffuse", colour=rng.uniform(0.2, 0.9, 3).astype(np.single)) if kind == 0 else
                        dict(kind="metal", colour=rng.uniform(0.6, 1, 3).astype(np.single), fuzziness=0.0) if kind == 1
                        else dict(kind="dielectric", ior=1.5))
        if k % 2 == 0:
            groups.append(dict(type="spheres", centres=c.copy(), radii=rad.copy(),
                               materials=np.full(n_per_group, m, np.ubyte)))
        else:
            groups.append(dict(type="triangles", pt0=p0.copy(), A=A.copy(), B=B.copy(), n0=nrm, n1=nrm, n2=nrm, uv0=uv,
                               uv1=uv, uv2=uv, material_id=m, bounds_centre=bc, bounds_radius=br))
    cam = numpy_path.camera_basis(np.array([5.0, 2.5, 6.0]), np.array([0.0, 0.9, 0.0]), 8.0, 0.0, 1.6, 45.0)
    sky = dict(horizon=np.array([1, 1, 1], np.single), zenith=np.array([0.5, 0.7, 1.0], np.single))
    return dict(camera=cam, groups=groups, materials=materials, sky=sky)


@pytest.mark.parametrize("n_copies,n_per_group", [(4, 12), (6, 40), (40, 10)])
def test_ties_between_large_groups(n_copies, n_per_group):
    from weekend_raytracer import main
    scene = layered_scene(n_copies, n_per_group, seed=n_copies)
    w, h, s, b = 48, 30, 2, 6
    want = numpy_path.render(scene, w, h, s, b, seed=77)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    got = main.render_frame(objs, w, h, s, b, seed=77, want_paths=True)
    assert np.array_equal(got["path_colours"], want["path_colours"])
    assert np.array_equal(got["segments"], want["segments"])
    assert np.array_equal(got["counts"], want["counts"])


def doubled_scene(n_per_group, seed, interleave):
    """Every sphere and every triangle twice inside its group, the copy with another material (spheres) or with
    other vertex normals and texture coordinates (triangles): every hit is a tie inside a group, which the lower
    index must win (numpy.argmin), however the wavefront's pooled lists order the candidates."""
    rng = np.random.default_rng(seed)
    base = layered_scene(1, n_per_group, seed)
    spheres = base["groups"][1]
    n = n_per_group
    order = np.arange(2 * n).reshape(2, n).T.reshape(-1) if interleave else np.arange(2 * n)
    materials = dict(base["materials"])
    materials[2] = dict(kind="metal", colour=np.array([0.9, 0.6, 0.2], np.single), fuzziness=0.0)
    materials[3] = dict(kind="diffuse", colour=np.array([0.2, 0.8, 0.3], np.single))
    sph = dict(type="spheres", centres=np.concatenate([spheres["centres"]] * 2)[order],
               radii=np.concatenate([spheres["radii"]] * 2)[order],
               materials=np.concatenate([np.full(n, 1, np.ubyte), np.full(n, 2, np.ubyte)])[order])
    p0 = (rng.uniform(-2, 2, (n, 3)) * np.array([1, 0.4, 1]) + np.array([0, 1.2, 0])).astype(np.single)
    A = rng.uniform(-0.8, 0.8, (n, 3)).astype(np.single)
    B = rng.uniform(-0.8, 0.8, (n, 3)).astype(np.single)
    nrm = np.cross(A, B)
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.single)
    tilted = nrm + np.array([0.3, 0.2, -0.1], np.single)
    tilted = (tilted / np.linalg.norm(tilted, axis=1)[:, None]).astype(np.single)
    two = lambda a, b: np.concatenate([a, b])[order]
    zero, one = np.zeros((n, 2), np.single), np.ones((n, 2), np.single)
    pts = np.concatenate([p0, p0 + A, p0 + B])
    bc = np.mean(pts, axis=0).astype(np.single)
    br = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
    tri = dict(type="triangles", pt0=two(p0, p0), A=two(A, A), B=two(B, B), n0=two(nrm, tilted), n1=two(nrm, tilted),
               n2=two(nrm, tilted), uv0=two(zero, one), uv1=two(zero, one), uv2=two(zero, one), material_id=3,
               bounds_centre=bc, bounds_radius=br)
    return dict(camera=base["camera"], groups=[base["groups"][0], sph, tri], materials=materials, sky=base["sky"])


@pytest.mark.parametrize("n_per_group,interleave", [(12, False), (12, True), (150, False), (150, True)])
def test_ties_inside_a_group(n_per_group, interleave):
    from weekend_raytracer import main
    scene = doubled_scene(n_per_group, 5 + n_per_group, interleave)
    w, h, s, b = 64, 40, 2, 6
    want = numpy_path.render(scene, w, h, s, b, seed=78)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    import torch
    for hierarchy in (0, 16):
        frame = main.Frame(objs, hierarchy=hierarchy)
        paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
        frame.accumulate(w, h, 0, s, b, 78, paths=paths)
        got = paths.cpu().numpy()
        frame.ctx.close()
        assert np.array_equal(got[:, :3], want["path_colours"]), hierarchy
