"""
Pins the CPU restatement (oracle/numpy_path.py) against the fixtures that
oracle/gen_golden.py produced by running the reference itself.  Everything
here is required to match BIT FOR BIT: the restatement evaluates the same
numpy operations on the same keyed random streams.
"""
import numpy as np
import pytest

from oracle import keyed_rng, numpy_path
from tests import helpers


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 with 10 rounds (the generator's default) and with 7 (the keyed streams')
    def run(c, k, rounds):
        return [int(v) for v in keyed_rng.philox4x32(*c, *k, rounds=rounds)]
    pi = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)
    assert run((0, 0, 0, 0), (0, 0), 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run((0xffffffff,) * 4, (0xffffffff,) * 2, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run(*pi, 10) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert keyed_rng.ROUNDS == 7
    assert run((0, 0, 0, 0), (0, 0), 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert run((0xffffffff,) * 4, (0xffffffff,) * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert run(*pi, 7) == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]


def test_known_answers_from_reference_tests():
    k = helpers.golden("kat")
    # /root/reference/tests/test_main.py:15-78
    for i in range(k["sph_o"].shape[0]):
        g = dict(type="spheres", centres=k["sph_c"][i:i + 1], radii=k["sph_r"][i:i + 1],
                 materials=np.array([7], dtype=np.ubyte))
        r = numpy_path.hit_spheres(g, k["sph_o"][i:i + 1], k["sph_d"][i:i + 1])
        assert r["hit"][0] and r["mat"][0] == 7
        assert abs(float(r["t"][0]) - float(k["sph_expected"][i])) < 1e-5
        assert r["t"][0] == k["sph_t"][i]
    # /root/reference/dev_helpers/sandbox.py:2302-2321
    tri = dict(type="triangles",
               pt0=np.array([[-1, -1, -3]], np.single), A=np.array([[2, 0, 0]], np.single),
               B=np.array([[1, 2, 0]], np.single),
               n0=np.array([[0, 1, 0]], np.single), n1=np.array([[0, 1, 0]], np.single),
               n2=np.array([[0, 1, 0]], np.single),
               uv0=np.array([[0, 0]], np.single), uv1=np.array([[1, 0]], np.single),
               uv2=np.array([[0.5, 1]], np.single), material_id=3,
               bounds_centre=np.array([0, -1 / 3, -3], np.single), bounds_radius=np.single(2.0))
    r = numpy_path.hit_triangles(tri, k["tri_o"], k["tri_d"])
    assert np.array_equal(r["hit"], k["tri_hit"]) and np.array_equal(r["t"], k["tri_t"])
    assert np.array_equal(r["p"], k["tri_p"]) and np.array_equal(r["n"], k["tri_n"])
    assert np.array_equal(r["uv"], k["tri_uv"]) and np.array_equal(r["back"], k["tri_back"])
    assert r["t"][0] == 3.0 and list(r["uv"][0]) == [0.5, 0.5] and list(r["back"]) == [False, True]
    inner = dict(type="spheres", centres=np.zeros((1, 3), np.single), radii=np.array([2.0], np.single),
                 materials=np.array([9], dtype=np.ubyte))
    r = numpy_path.hit_spheres(inner, k["in_o"], k["in_d"])
    assert r["t"][0] == 2.0 and r["back"][0] and r["mat"][0] == 9 and np.array_equal(r["n"], k["in_n"])


def test_camera_rays():
    c = helpers.golden("camera")
    seed = int(c["seed"])
    for key in c["names"]:
        focus, aperture, aspect, hfov = c[f"{key}_args"]
        w, h, s, s_begin = (int(v) for v in c[f"{key}_size"])
        cam = numpy_path.camera_basis(c[f"{key}_position"], c[f"{key}_lookat"], focus, aperture, aspect, hfov)
        for f in ("pos", "U", "V", "W", "horiz", "vert", "bottom_left", "lens_radius"):
            assert np.array_equal(np.asarray(cam[f], dtype=np.double), c[f"{key}_cam_{f}"]), (key, f)
        o, d = numpy_path.generate_rays(cam, w, h, s, seed, s_begin)
        assert np.array_equal(o, c[f"{key}_origins"].reshape(-1, 3)), key
        assert np.array_equal(d, c[f"{key}_dirs"].reshape(-1, 3)), key


@pytest.mark.parametrize("name", helpers.intersect_names())
def test_intersection(name):
    f = helpers.golden("intersect_" + name)
    scene = helpers.golden_scene(f)
    o, d = f["origins"], f["dirs"]
    for gi, g in enumerate(scene["groups"]):
        r = numpy_path.hit_group(g, o, d)
        for k in ("hit", "t", "p", "n", "uv", "mat", "back"):
            assert np.array_equal(r[k], f[f"g{gi}_{k}"]), (name, gi, k)
        if f"g{gi}_prim" in f:
            assert np.array_equal(r["prim"], f[f"g{gi}_prim"]), (name, gi)
    m = numpy_path.closest_hit(scene["groups"], o, d)
    for k in ("hit", "t", "p", "n", "uv", "mat", "back", "group", "prim"):
        assert np.array_equal(m[k], f[f"merged_{k}"]), (name, k)


def test_scatter_all_materials():
    f = helpers.golden("scatter")
    scene = helpers.golden_scene(f)
    seed = int(f["seed"])
    ident = (f["px"], f["py"], f["ps"])
    for k, name in enumerate(f["names"]):
        name = str(name)
        d, col, ab = numpy_path.scatter(scene["materials"][k], f["dirs"], f["points"], f["normals"], f["uvs"],
                                        f["back"], seed, ident, int(f[name + "_bounce"]))
        assert np.array_equal(np.asarray(d, dtype=np.double), f[name + "_dirs64"]), name
        assert np.array_equal(np.asarray(col).astype(np.single), f[name + "_colours"]), name
        assert np.array_equal(ab, f[name + "_absorbed"]), name
    # the interesting branches are really present in the fixture
    assert f["metal_fuzzy_absorbed"].any() and not f["metal_fuzzy_absorbed"].all()
    glass = f["dielectric_dirs64"]
    refl = numpy_path._reflect(f["dirs"], f["normals"])
    is_reflect = np.all(np.abs(glass - refl.astype(np.single)) < 1e-6, axis=1)
    assert is_reflect.any() and not is_reflect.all()


@pytest.mark.parametrize("name", helpers.frame_names())
def test_whole_frame(name):
    f = helpers.golden("frame_" + name)
    scene = helpers.golden_scene(f)
    w, h, s, b, s_begin = (int(v) for v in f["size"])
    if w * h * s * sum(g["A"].shape[0] for g in scene["groups"] if g["type"] == "triangles") > 6e7:
        pytest.skip("too slow for the CPU suite; covered by the GPU parity test")
    trace = []
    r = numpy_path.render(scene, w, h, s, b, int(f["seed"]), s_begin, trace=trace)
    assert np.array_equal(r["counts"], f["counts"])
    assert np.array_equal(r["segments"], f["segments"])
    assert np.array_equal(r["path_colours"], f["path_colours"])
    assert np.array_equal(r["sample_sum"], f["sample_sum"])
    assert np.array_equal(r["image"], f["image"])
    if "trace_b0_active" in f:
        for rec in trace:
            bn = rec["bounce"]
            assert np.array_equal(rec["active"], f[f"trace_b{bn}_active"])
            for k in ("hit", "t", "p", "n", "uv", "mat", "back", "group"):
                assert np.array_equal(rec["rec"][k], f[f"trace_b{bn}_{k}"]), (bn, k)
            for k in ("out_o", "out_d", "colour", "absorbed"):
                assert np.array_equal(rec[k], f[f"trace_b{bn}_{k}"]), (bn, k)


def test_unit_sphere_candidates_are_statistically_the_references_draw():
    """The keyed unit-sphere candidates lie on a 21-bit lattice, -1 + v * 2**-20, where the reference
    draws a float64 uniform of [-1, 1) and casts it to float32 (materials.py:407-408).  The lattice's
    distribution must be the same up to its resolution: acceptance rate of the rejection test
    (materials.py:417-420: pi / 6), first and second moments of the accepted and normalised
    directions, against numpy's own draw of the same size (differences within sampling noise), and the
    lattice's asymmetry, a mean of -2**-21 per coordinate, far below it."""
    from oracle import keyed_rng
    n = 1_000_000
    ids = np.arange(n)
    x, y, s = ids % 1024, (ids // 1024) % 1024, ids // (1024 * 1024)
    ours = np.concatenate([keyed_rng.unit_candidate(99, x, y, s, 3, attempt).astype(np.single) for attempt in (0, 1)])
    theirs = np.random.default_rng(99).uniform(-1.0, 1.0, size=ours.shape).astype(np.single)
    # every value is on the lattice and exact in float32
    assert np.all((ours.astype(np.float64) + 1.0) * 1048576.0 % 1.0 == 0.0)
    assert ours.min() >= -1.0 and ours.max() <= 1.0 - 2.0 ** -20
    m = ours.shape[0]
    noise = 4.0 / np.sqrt(m)                                     # four standard errors of a mean of values of [-1, 1]

    def summary(c):
        len2 = np.einsum("ij,ij->i", c, c)
        ok = ~np.logical_or(len2 > 1.0, len2 < 0.00001)
        unit = c[ok] / np.sqrt(len2[ok])[:, None]
        return ok.mean(), c.mean(axis=0), (c.astype(np.float64) ** 2).mean(axis=0), unit.mean(axis=0), \
            (unit[:, :, None].astype(np.float64) * unit[:, None, :]).mean(axis=0)

    a_ok, a_mean, a_sq, a_dir, a_cov = summary(ours)
    b_ok, b_mean, b_sq, b_dir, b_cov = summary(theirs)
    assert abs(a_ok - np.pi / 6.0) < noise and abs(a_ok - b_ok) < 2 * noise
    assert np.all(np.abs(a_mean) < noise) and np.all(np.abs(a_sq - 1.0 / 3.0) < noise)
    assert np.all(np.abs(a_dir) < 2 * noise) and np.all(np.abs(a_dir - b_dir) < 3 * noise)
    assert np.all(np.abs(a_cov - np.eye(3) / 3.0) < 2 * noise) and np.all(np.abs(a_cov - b_cov) < 3 * noise)
    # the two candidates of one generator call are not correlated with each other
    even, odd = ours[:n].astype(np.float64), ours[n:].astype(np.float64)
    assert np.all(np.abs((even[:, :, None] * odd[:, None, :]).mean(axis=0)) < noise)
