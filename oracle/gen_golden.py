#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE - golden-vector generator.  Runs ONLY in the build
container (it imports the reference from /root/reference); nothing here is
imported by the shipped renderer, and only the .npz files it writes travel.

What it does
  1. imports the reference package with a stub `humanize` module (used by the
     reference only inside print statements: sphere_group.py:60-62,173,
     triangle_group.py:137-139,323);
  2. replaces the reference's three random sources with keyed streams
     (oracle/keyed_rng.py): camera.RNG, materials.RNG and
     materials.numpy_random_unit_vecs;
  3. drives the reference's own Camera.get_ray_components / *.get_hits /
     *.scatter from a restatement of the bounce loop of main.render()
     (main.py:93-220) that knows which path every ray belongs to;
  4. proves that restatement equal, bit for bit, to the unmodified
     main.render() under ordinary seeded sequential generators;
  5. writes inputs + the reference's outputs to tests/golden/*.npz.

Documented deviation: Metal.scatter with fuzziness > 1e-4 raises NameError in
the reference as shipped (materials.py:273-276).  `metal_scatter_intended`
below restates that method with the two garbled lines repaired; it is
installed for every run, and is byte-for-byte the shipped behaviour whenever
fuzziness <= 1e-4.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import keyed_rng, scene_io  # noqa: E402

REF_ROOT = "/root/reference"
GOLDEN = os.path.join(REPO, "tests", "golden")


# --------------------------------------------------------------------------
# reference import + shims
# --------------------------------------------------------------------------
def import_reference():
    stub = types.ModuleType("humanize")
    stub.naturalsize = lambda value, binary=False: str(value)
    sys.modules["humanize"] = stub
    sys.path.insert(0, os.path.join(REF_ROOT, "src"))
    os.chdir(os.path.join(REF_ROOT, "dev_helpers"))
    import weekend_raytracer
    from weekend_raytracer import (camera, disk, main, materials, obj_tri_mesh,
                                   scenes, sphere_group, triangle_group)
    ref = types.SimpleNamespace(
        camera=camera, disk=disk, main=main, materials=materials,
        obj_tri_mesh=obj_tri_mesh, scenes=scenes, sphere_group=sphere_group,
        triangle_group=triangle_group,
    )
    assert weekend_raytracer.__file__.startswith(REF_ROOT)

    def metal_scatter_intended(self, hit_raydirs, hit_points, hit_normals, hit_uvs, hit_backfaces):
        reflected_dirs = materials.numpy_reflect(hit_raydirs, hit_normals)
        hit_cols = numpy.full((hit_points.shape[0], 3), self.colour, dtype=numpy.single)
        absorbtions = numpy.full((hit_points.shape[0]), False)
        if self.fuzziness > 0.0001:
            fuzz_offsets = materials.numpy_random_unit_vecs(hit_points.shape[0]) * self.fuzziness
            reflected_dirs += fuzz_offsets
            reflected_dirs /= numpy.sqrt(
                numpy.einsum("ij,ij->i", reflected_dirs, reflected_dirs))[..., numpy.newaxis]
            cos_angles = numpy.einsum("ij,ij->i", reflected_dirs, hit_normals)
            scattered_inside = cos_angles < 0.00001
            hit_cols[scattered_inside] = 0.0
            absorbtions[scattered_inside] = True
        return hit_points, reflected_dirs, hit_cols, absorbtions

    materials.Metal.scatter = metal_scatter_intended
    return ref


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


class NumpyProxy:
    """Stands in for the `numpy` name inside one reference module."""

    def __init__(self, full_fix=False):
        self.full_fix = full_fix
        self.last_argmin = None

    def __getattr__(self, name):
        return getattr(numpy, name)

    def full(self, shape, fill_value, dtype=None, **kw):
        # numpy >= 2 refuses full(..., -1, dtype=ubyte) (main.py:121); numpy 1
        # wrapped it to 255, which is also what sphere_group.py:255-259 yields.
        if self.full_fix and dtype is numpy.ubyte and fill_value == -1:
            fill_value = 255
        return numpy.full(shape, fill_value, dtype=dtype, **kw)

    def argmin(self, a, axis=None, **kw):
        self.last_argmin = numpy.argmin(a, axis=axis, **kw)
        return self.last_argmin


# --------------------------------------------------------------------------
# keyed stream injection
# --------------------------------------------------------------------------
class KeyedCameraRNG:
    """camera.RNG stand-in.  Call 0 is the jitter draw (camera.py:130), call k>0
    is lens-disk attempt k-1 (camera.py:138,145).  Entry [x,y,s,:] is keyed by
    pixel (x,y) and global sample s_begin+s."""

    def __init__(self, seed, s_begin=0):
        self.seed = seed
        self.s_begin = s_begin
        self.calls = 0

    def uniform(self, low=0.0, high=1.0, size=None):
        width, height, samples, three = size
        assert three == 3
        x, y, s = numpy.meshgrid(numpy.arange(width), numpy.arange(height),
                                 numpy.arange(samples) + self.s_begin, indexing="ij")
        out = numpy.zeros(size)            # the third component is never used (camera.py:131,140)
        if self.calls == 0:
            assert (low, high) == (0.0, 1.0)
            out[..., 0:2] = keyed_rng.jitter(self.seed, x, y, s)
        else:
            assert (low, high) == (-1.0, 1.0)
            out[..., 0:2] = keyed_rng.lens_candidate(self.seed, x, y, s, self.calls - 1)
        self.calls += 1
        return out


class ScatterContext:
    """Tells the injected material streams which paths are being scattered."""

    def __init__(self, seed):
        self.seed = seed
        self.x = self.y = self.s = None
        self.bounce = 0

    def set(self, x, y, s, bounce):
        self.x, self.y, self.s, self.bounce = x, y, s, bounce

    # materials.numpy_random_unit_vecs stand-in
    def unit_vecs(self, num_vecs):
        assert num_vecs == self.x.shape[0]
        return keyed_rng.keyed_unit_vectors(self.seed, self.x, self.y, self.s, self.bounce)

    # materials.RNG stand-in (Dielectric.scatter, materials.py:331)
    def uniform(self, low=0.0, high=1.0, size=None):
        n = size if isinstance(size, int) else size[0]
        assert n == self.x.shape[0]
        return low + (high - low) * keyed_rng.keyed_schlick(self.seed, self.x, self.y, self.s, self.bounce)


def path_identity(width, height, samples, s_begin=0):
    ids = numpy.arange(width * height * samples)
    return ids // (height * samples), (ids // samples) % height, ids % samples + s_begin


# --------------------------------------------------------------------------
# restated bounce loop (main.py:93-220)
# --------------------------------------------------------------------------
def merge_hits(object_groups, origins, dirs, t_min=0.001, t_max=1000.0):
    """main.py:115-173 for one set of rays; also reports which group won."""
    n = origins.shape[0]
    ray_hits = numpy.full((n,), False)
    hit_ts = numpy.full((n,), 5001.0)
    hit_pts = numpy.full((n, 3), 0.0)
    hit_normals = numpy.full((n, 3), 0.0)
    hit_uvs = numpy.full((n, 2), 0.0)
    hit_material_indecies = numpy.full((n,), 255, dtype=numpy.ubyte)
    back_facing = numpy.full((n,), False)
    hit_group = numpy.full((n,), -1, dtype=numpy.int32)
    for group_index, object_group in enumerate(object_groups):
        (tmp_ray_hits, tmp_hit_ts, tmp_hit_pts, tmp_hit_normals, tmp_hit_uvs,
         tmp_hit_material_indecies, tmp_back_facing) = object_group.get_hits(origins, dirs, t_min, t_max)
        ray_hits = ray_hits | tmp_ray_hits
        condition = tmp_ray_hits & (tmp_hit_ts < hit_ts)
        hit_ts = numpy.where(condition, tmp_hit_ts, hit_ts)
        hit_pts = numpy.where(condition[..., numpy.newaxis], tmp_hit_pts, hit_pts)
        hit_normals = numpy.where(condition[..., numpy.newaxis], tmp_hit_normals, hit_normals)
        hit_uvs = numpy.where(condition[..., numpy.newaxis], tmp_hit_uvs, hit_uvs)
        hit_material_indecies = numpy.where(condition, tmp_hit_material_indecies, hit_material_indecies)
        back_facing = numpy.where(condition, tmp_back_facing, back_facing)
        hit_group = numpy.where(condition, group_index, hit_group)
    return ray_hits, hit_ts, hit_pts, hit_normals, hit_uvs, hit_material_indecies, back_facing, hit_group


def harness_render(ref, camera, object_groups, material_map, width, height, samples, max_bounces,
                   seed=None, s_begin=0, trace=None):
    """seed=None: use whatever generators are installed in the reference
    modules (the self-check); otherwise install keyed streams."""
    main = ref.main
    px, py, ps = path_identity(width, height, samples, s_begin)
    saved = (ref.camera.RNG, ref.materials.RNG, ref.materials.numpy_random_unit_vecs)
    ctx = None
    if seed is not None:
        ctx = ScatterContext(seed)
        ref.camera.RNG = KeyedCameraRNG(seed, s_begin)
        ref.materials.RNG = ctx
        ref.materials.numpy_random_unit_vecs = ctx.unit_vecs
    try:
        ray_colours = numpy.ones((width, height, samples, max_bounces + 1, 3), dtype=numpy.single)
        ray_origins, ray_directions = camera.get_ray_components(width, height, samples)
        ray_origins = ray_origins.reshape(-1, 3)
        ray_directions = ray_directions.reshape(-1, 3)
        ray_colours = ray_colours.reshape(-1, max_bounces + 1, 3)
        active = numpy.arange(ray_origins.shape[0])
        counts = []
        segments = numpy.zeros(ray_origins.shape[0], dtype=numpy.int32)
        bounce = 0
        while bounce <= max_bounces:
            counts.append(active.shape[0])
            if bounce != max_bounces:
                segments[active] += 1
                in_o = ray_origins[active]
                in_d = ray_directions[active]
                (ray_hits, hit_ts, hit_pts, hit_normals, hit_uvs, hit_mats, back_facing,
                 hit_group) = merge_hits(object_groups, in_o, in_d)
                ray_misses = numpy.logical_not(ray_hits)
                absorbs = numpy.full((ray_hits.shape[0]), False)
                for material_index, material in material_map.items():
                    matches = (hit_mats == material_index) & ray_hits
                    sel = active[matches]
                    if ctx is not None:
                        ctx.set(px[sel], py[sel], ps[sel], bounce)
                    s_o, s_d, s_c, s_a = material.scatter(
                        ray_directions[sel], hit_pts[matches], hit_normals[matches],
                        hit_uvs[matches], back_facing[matches])
                    ray_origins[sel] = s_o
                    ray_directions[sel] = s_d
                    ray_colours[sel, bounce] = s_c
                    absorbs[matches] = s_a
                miss_sel = active[ray_misses]
                ts = (ray_directions[miss_sel, 1] + 1.0) * 0.5
                ray_colours[miss_sel, bounce] = (
                    (1.0 - ts)[..., numpy.newaxis] * main.HORIZON_COLOUR
                    + ts[..., numpy.newaxis] * main.SKY_COLOUR)
                if trace is not None:
                    trace.append(dict(
                        bounce=bounce, active=active.copy(), in_o=in_o, in_d=in_d,
                        hit=ray_hits, t=hit_ts, p=hit_pts, n=hit_normals, uv=hit_uvs,
                        mat=hit_mats, back=back_facing, group=hit_group,
                        out_o=ray_origins[active].copy(), out_d=ray_directions[active].copy(),
                        colour=ray_colours[active, bounce].copy(), absorbed=absorbs.copy()))
                active = active[numpy.logical_and(ray_hits, numpy.logical_not(absorbs))]
            else:
                ray_colours[active, bounce] = 0.0
            bounce += 1
        path_colours = numpy.prod(ray_colours, axis=1)
        stacked = path_colours.reshape(width, height, samples, 3)
        image = numpy.sqrt(numpy.mean(stacked, axis=2))
        return dict(image=image, path_colours=path_colours, counts=numpy.array(counts, dtype=numpy.int64),
                    segments=segments, sample_sum=numpy.sum(stacked, axis=2, dtype=numpy.single))
    finally:
        ref.camera.RNG, ref.materials.RNG, ref.materials.numpy_random_unit_vecs = saved


def self_check(ref):
    """The restated loop must equal the unmodified main.render() bit for bit
    when both consume the same ordinary sequential generators."""
    main = ref.main
    for scene_name, (w, h, s, b) in {
        "numpy_one_weekend_demo_scene": (40, 22, 3, 5),
        "numpy_simple_sphere_scene": (32, 18, 2, 4),
        "numpy_triangles_scene": (24, 14, 2, 3),
        "disk_test_scene": (24, 14, 2, 3),
    }.items():
        scene_fn = getattr(ref.scenes, scene_name)
        aspect = w / h

        def seed_all():
            ref.camera.RNG = numpy.random.default_rng(1001)
            ref.materials.RNG = numpy.random.default_rng(1002)

        saved_scene = ref.scenes.ray_group_triangle_group_bunny_scene
        saved_np = main.numpy
        try:
            ref.scenes.ray_group_triangle_group_bunny_scene = scene_fn
            main.numpy = NumpyProxy(full_fix=True)
            main.IMG_WIDTH, main.IMG_HEIGHT, main.PIXEL_SAMPLES, main.MAX_BOUNCES = w, h, s, b
            main.ASPECT_RATIO = aspect
            seed_all()
            with quiet():
                pixels = main.render()
        finally:
            ref.scenes.ray_group_triangle_group_bunny_scene = saved_scene
            main.numpy = saved_np
        expected = numpy.zeros((w, h, 3), dtype=numpy.single)
        for (x, y), rgb in pixels.items():
            expected[x, y] = rgb
        seed_all()
        with quiet():
            cam, groups, mats = scene_fn(aspect)
            got = harness_render(ref, cam, groups, mats, w, h, s, b, seed=None)
        if not numpy.array_equal(got["image"], expected):
            raise SystemExit(f"self-check FAILED for {scene_name}")
        print(f"self-check ok: {scene_name} {w}x{h}x{s} b{b}: restated loop == main.render() bit for bit")


# --------------------------------------------------------------------------
# reference objects -> flat scene
# --------------------------------------------------------------------------
def flatten_scene(ref, camera, object_groups, material_map):
    cam = dict(pos=camera.camera_pos, U=camera.U, V=camera.V, W=camera.W,
               horiz=camera.viewport_horizontal, vert=camera.viewport_vertical,
               bottom_left=camera.bottomleft_focalplane_pos, lens_radius=float(camera.lens_radius))
    groups = []
    for g in object_groups:
        if isinstance(g, ref.sphere_group.SphereGroup):
            groups.append(dict(type="spheres", centres=g.centres, radii=g.radii,
                               materials=g.material_indecies))
        elif isinstance(g, ref.triangle_group.TriangleGroup):
            groups.append(dict(type="triangles", pt0=g.pt0s, A=g.As, B=g.Bs,
                               n0=g.normal0s, n1=g.normal1s, n2=g.normal2s,
                               uv0=g.uv0s, uv1=g.uv1s, uv2=g.uv2s, material_id=int(g.material_id),
                               bounds_centre=g.bounds_centre, bounds_radius=g.bounds_radius))
        elif isinstance(g, ref.disk.Disk):
            groups.append(dict(type="disk", centre=numpy.asarray(g.centre), normal=g.normal, U=g.U, V=g.V,
                               radius=float(g.radius), material_index=int(g.material_index)))
        else:
            raise TypeError(type(g))
    mats = {}
    M = ref.materials
    for index, m in material_map.items():
        if isinstance(m, M.Diffuse):
            mats[index] = dict(kind="diffuse", colour=numpy.asarray(m.colour))
        elif isinstance(m, M.TexturedDiffuse):
            mats[index] = dict(kind="textured", pixels=numpy.ascontiguousarray(m.texture_pixels),
                               smallest_side=float(m.smallest_side))
        elif isinstance(m, M.CheckerboardDiffuse):
            mats[index] = dict(kind="checker", scale=numpy.asarray(m.scale), offset=numpy.asarray(m.offset),
                               colour_a=numpy.asarray(m.colour_a), colour_b=numpy.asarray(m.colour_b))
        elif isinstance(m, M.NormalToRGBDiffuse):
            mats[index] = dict(kind="normal_rgb")
        elif isinstance(m, M.NormalToDiscreteRGBDiffuse):
            mats[index] = dict(kind="normal_discrete")
        elif isinstance(m, M.Metal):
            mats[index] = dict(kind="metal", colour=numpy.asarray(m.colour), fuzziness=float(m.fuzziness))
        elif isinstance(m, M.Dielectric):
            mats[index] = dict(kind="dielectric", ior=float(m.ior))
        else:
            raise TypeError(type(m))
    sky = dict(horizon=ref.main.HORIZON_COLOUR, zenith=ref.main.SKY_COLOUR)
    return dict(camera=cam, groups=groups, materials=mats, sky=sky)


# --------------------------------------------------------------------------
# scenes the reference does not ship as functions (built from its classes)
# --------------------------------------------------------------------------
def sphere_field_records():
    """The shipped small-sphere field (src/weekend_raytracer/sphere_data.json,
    187 records) that scenes.py:336-357 would load."""
    with open(os.path.join(REF_ROOT, "src", "weekend_raytracer", "sphere_data.json")) as fh:
        return json.load(fh)


def cover_scene(ref, aspect_ratio, records, aperture=0.1):
    """Book-1 cover: the demo scene's camera with the depth-of-field aperture
    of scenes.py:280, ground + three unit spheres (lambertian brown / glass /
    metal as archived_scenes.py:141-178 intended) + one material per small
    sphere (uint8 material indices: at most 255 distinct materials)."""
    M = ref.materials
    cam = ref.camera.Camera(numpy.array([13.0, 2.0, 3.0]), numpy.array([0.0, 0.5, 0.0]),
                            10.0, aperture, aspect_ratio, 30.0)
    material_map = {
        0: M.Diffuse(numpy.array([0.5, 0.5, 0.5], dtype=numpy.single)),
        1: M.Diffuse(numpy.array([0.4, 0.2, 0.1], dtype=numpy.single)),
        2: M.Dielectric(1.5),
        3: M.Metal(numpy.array([0.7, 0.6, 0.5], dtype=numpy.single), 0.0),
    }
    grp = ref.sphere_group.SphereGroup()
    black = numpy.array([0, 0, 0], dtype=numpy.single)
    grp.add_sphere(numpy.array([0, -1000, 0], dtype=numpy.single), 1000.0, black, 0)
    grp.add_sphere(numpy.array([-4, 1, 0], dtype=numpy.single), 1.0, black, 1)
    grp.add_sphere(numpy.array([0, 1, 0], dtype=numpy.single), 1.0, black, 2)
    grp.add_sphere(numpy.array([4, 1, 0], dtype=numpy.single), 1.0, black, 3)
    next_index = 4
    for rec in records:
        if rec["material"] == "diffuse":
            mat = M.Diffuse(numpy.array(rec["colour"], dtype=numpy.single))
        elif rec["material"] == "metal":
            mat = M.Metal(numpy.array(rec["colour"], dtype=numpy.single), rec["fuzziness"])
        else:
            mat = M.Dielectric(rec["ior"])
        if next_index > 254:
            raise ValueError("more than 255 materials do not fit the reference's uint8 indices")
        material_map[next_index] = mat
        grp.add_sphere(numpy.array(rec["pos"], dtype=numpy.single), rec["radius"], black, next_index)
        next_index += 1
    return cam, [grp], material_map


def mesh_group(ref, obj_name, material_id, smooth=False, use_uvs=True, lift=True):
    """A TriangleGroup filled the way scenes.py:612-640 fills one."""
    mesh = ref.obj_tri_mesh.OBJTriMesh()
    mesh.read(obj_name)
    smallest_y = min(v[1] for v in mesh.vertices) if lift else 0.0
    grp = ref.triangle_group.TriangleGroup(material_id)
    for tri in mesh.faces:
        pts = [numpy.array([mesh.vertices[tri[k][0]][0], mesh.vertices[tri[k][0]][1] - smallest_y,
                            mesh.vertices[tri[k][0]][2]]) for k in range(3)]
        kw = {}
        if use_uvs:
            kw.update(uv0=numpy.array(mesh.uvs[tri[0][1]]), uv1=numpy.array(mesh.uvs[tri[1][1]]),
                      uv2=numpy.array(mesh.uvs[tri[2][1]]))
        if smooth:
            kw.update(normal0=mesh.get_smooth_vertex_normal(tri[0][0]),
                      normal1=mesh.get_smooth_vertex_normal(tri[1][0]),
                      normal2=mesh.get_smooth_vertex_normal(tri[2][0]))
        grp.add_triangle(*pts, **kw)
    return grp


def cloth_bunny_scene(ref, aspect_ratio):
    """The textured-bunny-and-cloth benchmark configuration without strength maps (the reference has no such
    material): the shipped textured bunny scene (scenes.py:563-661) plus
    smallCloth.obj UV-mapped with picnicCloth.jpg."""
    cam, groups, material_map = ref.scenes.ray_group_triangle_group_bunny_scene(aspect_ratio)
    material_map[3] = ref.materials.TexturedDiffuse("picnicCloth.jpg")
    cloth = mesh_group(ref, "smallCloth.obj", 3, smooth=False, use_uvs=True, lift=False)
    return cam, [groups[0], cloth, groups[1]], material_map


def bunny_diffuse_scene(ref, aspect_ratio):
    """The mesh benchmark configuration: the bunny scene with the texture swapped for the
    Lambertian `bunny_mat` the scene defines but leaves unused (scenes.py:584-586)."""
    cam, groups, material_map = ref.scenes.ray_group_triangle_group_bunny_scene(aspect_ratio)
    material_map[1] = ref.materials.Diffuse(numpy.array([0.2, 0.7, 0.1], dtype=numpy.single))
    return cam, groups, material_map


def placed_mesh(ref, obj_name, material_id, offset):
    """A TriangleGroup filled the way the four-bunnies and cow scenes fill theirs (scenes.py:57-190,
    236-271): vertices lifted to y = 0 and shifted in x and z as Python floats, points made float32."""
    mesh = ref.obj_tri_mesh.OBJTriMesh()
    mesh.read(obj_name)
    smallest_y = min(v[1] for v in mesh.vertices)
    grp = ref.triangle_group.TriangleGroup(material_id)
    for tri in mesh.faces:
        grp.add_triangle(*[numpy.array([mesh.vertices[tri[k][0]][0] + offset[0],
                                        mesh.vertices[tri[k][0]][1] - smallest_y,
                                        mesh.vertices[tri[k][0]][2] + offset[1]], dtype=numpy.single)
                           for k in range(3)])
    return grp


def cow_teapot_field_scene(ref, aspect_ratio, records):
    """The cow + teapot + sphere-field benchmark configuration (no reference scene uses teapot.obj):
    the cover scene's spheres and materials without depth of field, the cow scene's mesh material on
    the cow (scenes.py:196-273), a mirror teapot, both meshes built by the reference's own
    OBJTriMesh / TriangleGroup.add_triangle."""
    M = ref.materials
    _, (spheres,), material_map = cover_scene(ref, aspect_ratio, records)
    cam = ref.camera.Camera(numpy.array([13.0, 6.0, 9.0]), numpy.array([0.0, 1.5, 0.0]), 10.0, 0.0, aspect_ratio, 40.0)
    material_map[250] = M.CheckerboardDiffuse(numpy.array([1.0, 1.0, 1.0]), numpy.array([0.2, 0.2, 0.2]),
                                              numpy.array([0.8, 0.8, 0.85], dtype=numpy.single),
                                              numpy.array([0.25, 0.25, 0.2], dtype=numpy.single))
    material_map[251] = M.Metal(numpy.array([0.85, 0.85, 0.85], dtype=numpy.single), 0.0)
    cow = placed_mesh(ref, "cow.obj", 250, (-3.0, -2.5))
    teapot = placed_mesh(ref, "teapot.obj", 251, (3.0, 3.5))
    return cam, [spheres, cow, teapot], material_map


def write_smooth_normals(ref, obj_name, every):
    """OBJTriMesh.get_smooth_vertex_normal (obj_tri_mesh.py:27-68) of every `every`-th vertex of a mesh."""
    with quiet():
        mesh = ref.obj_tri_mesh.OBJTriMesh()
        mesh.read(obj_name)
        picked = numpy.arange(0, len(mesh.vertices), every)
        normals = numpy.array([mesh.get_smooth_vertex_normal(int(v)) for v in picked])
    save("smooth_normals_" + obj_name.split(".")[0], mesh=numpy.array(obj_name), vertices=picked, normals=normals,
         dtype=numpy.array(str(normals.dtype)))


# --------------------------------------------------------------------------
# fixture writers
# --------------------------------------------------------------------------
TEXTURE_FILES = {"bunnyTexture.tif", "picnicCloth.jpg", "uv_test.jpg"}


def texture_file_map(ref, material_map, names):
    out = {}
    tex = [i for i, m in material_map.items() if isinstance(m, ref.materials.TexturedDiffuse)]
    for i, name in zip(tex, names):
        out[i] = name
    return out


def save(name, **arrays):
    path = os.path.join(GOLDEN, name + ".npz")
    numpy.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")


def write_kat(ref):
    """Known answers salvaged from the reference's own (stale) tests and
    scratchpad, evaluated by the vectorised reference code."""
    # tests/test_main.py:15-78: (origin, direction, centre, radius, expected t)
    r2, r3, r5 = numpy.sqrt(2.0), numpy.sqrt(3.0), numpy.sqrt(5.0)
    cases = [
        ((0, 0, 0), (1, 0, 0), (3, 0, 0), 1.0, 2.0),
        ((0, 0, 0), (1, 1, 0), (3, 3, 0), 1.0, 3 * r2 - 1),
        ((0, 0, 0), (-1, -1, 0), (-3, -3, 0), 1.0, 3 * r2 - 1),
        ((0, 0, 5), (1, 1, 0), (3, 3, 5), 1.0, 3 * r2 - 1),
        ((0, 0, 0), (1, 1, 1), (3, 3, 3), 1.0, 3 * r3 - 1),
        ((-1, 2, 3), (1, 1, 1), (2, 5, 6), 1.0, 3 * r3 - 1),
        ((0, 0, 0), (1, 2, 0), (3, 5, 0), r2, 2 * r5),
        ((0, 0, 0), (1, 2, 0), (4, 6, 0), 2 * r2, 2 * r5),
    ]
    o = numpy.array([c[0] for c in cases], dtype=numpy.single)
    d = numpy.array([c[1] for c in cases], dtype=numpy.double)
    d = (d / numpy.linalg.norm(d, axis=1)[:, None]).astype(numpy.single)
    ts = []
    for i, c in enumerate(cases):
        grp = ref.sphere_group.SphereGroup()
        grp.add_sphere(numpy.array(c[2], dtype=numpy.single), c[3], numpy.zeros(3, numpy.single), 7)
        with quiet():
            hit, t, p, n, uv, mat, back = grp.get_hits(o[i:i + 1], d[i:i + 1], 0.001, 1000.0)
        assert hit[0] and mat[0] == 7 and abs(t[0] - c[4]) < 1e-5, (i, t, c[4])
        ts.append(t[0])
    # sandbox.py:2302-2321 triangle with UVs, seen from both sides
    tri = ref.triangle_group.TriangleGroup(3)
    tri.add_triangle(numpy.array([-1, -1, -3], dtype=numpy.single), numpy.array([1, -1, -3], dtype=numpy.single),
                     numpy.array([0, 1, -3], dtype=numpy.single),
                     uv0=numpy.array([0, 0], dtype=numpy.single), uv1=numpy.array([1, 0], dtype=numpy.single),
                     uv2=numpy.array([0.5, 1], dtype=numpy.single))
    to = numpy.array([[0, 0, 0], [0, 0, -6]], dtype=numpy.single)
    td = numpy.array([[0, 0, -1], [0, 0, 1]], dtype=numpy.single)
    with quiet():
        th = tri.get_hits(to, td, 0.001, 1000.0)
    # sphere seen from inside
    inner = ref.sphere_group.SphereGroup()
    inner.add_sphere(numpy.array([0, 0, 0], dtype=numpy.single), 2.0, numpy.zeros(3, numpy.single), 9)
    io_ = numpy.array([[0, 0, 0]], dtype=numpy.single)
    id_ = numpy.array([[0, 0, -1]], dtype=numpy.single)
    with quiet():
        ih = inner.get_hits(io_, id_, 0.001, 1000.0)
    save("kat",
         sph_o=o, sph_d=d, sph_c=numpy.array([c[2] for c in cases], dtype=numpy.single),
         sph_r=numpy.array([c[3] for c in cases], dtype=numpy.single),
         sph_expected=numpy.array([c[4] for c in cases]), sph_t=numpy.array(ts, dtype=numpy.single),
         tri_o=to, tri_d=td, tri_hit=th[0], tri_t=numpy.asarray(th[1], dtype=numpy.double), tri_p=th[2], tri_n=th[3],
         tri_uv=th[4], tri_mat=th[5], tri_back=th[6],
         in_o=io_, in_d=id_, in_hit=ih[0], in_t=ih[1], in_p=ih[2], in_n=ih[3], in_mat=ih[5], in_back=ih[6])


def write_camera(ref):
    out = {}
    cases = {
        "demo": dict(pos=numpy.array([13.0, 2.0, 3.0]), lookat=numpy.array([0.0, 0.5, 0.0]), focus=10.0,
                     aperture=0.0, hfov=30.0),
        "cover": dict(pos=numpy.array([13.0, 2.0, 3.0]), lookat=numpy.array([0.0, 0.5, 0.0]), focus=10.0,
                      aperture=0.1, hfov=30.0),
        "bunny": dict(pos=numpy.array([-2.0, 3.5, 8.0]), lookat=numpy.array([-2.2, 1.7, 0.0]), focus=10,
                      aperture=0.0, hfov=53.0),
        # scenes.py:364-368 passes float32 vectors
        "glass32": dict(pos=numpy.array([0, 0, 0], dtype=numpy.single),
                        lookat=numpy.array([0.0, 0.0, -5.0], dtype=numpy.single), focus=10.0, aperture=0.0,
                        hfov=90.0),
        "wide_lens": dict(pos=numpy.array([1.0, 2.0, 3.0]), lookat=numpy.array([0.0, 0.0, -1.0]), focus=4.5,
                          aperture=2.0, hfov=75.0),
    }
    seed = 4242
    names = []
    for name, c in cases.items():
        for (w, h, s, s_begin) in [(4, 2, 1, 0), (16, 9, 2, 0), (16, 9, 2, 6)]:
            cam = ref.camera.Camera(c["pos"], c["lookat"], c["focus"], c["aperture"], w / h, c["hfov"])
            saved = ref.camera.RNG
            ref.camera.RNG = KeyedCameraRNG(seed, s_begin)
            try:
                o, d = cam.get_ray_components(w, h, s)
            finally:
                ref.camera.RNG = saved
            key = f"{name}_{w}x{h}x{s}_s{s_begin}"
            names.append(key)
            out[key + "_args"] = numpy.array([c["focus"], c["aperture"], w / h, c["hfov"]], dtype=numpy.double)
            out[key + "_position"] = c["pos"]
            out[key + "_lookat"] = c["lookat"]
            out[key + "_size"] = numpy.array([w, h, s, s_begin])
            for f, v in dict(pos=cam.camera_pos, U=cam.U, V=cam.V, W=cam.W, horiz=cam.viewport_horizontal,
                             vert=cam.viewport_vertical, bottom_left=cam.bottomleft_focalplane_pos,
                             lens_radius=cam.lens_radius).items():
                out[f"{key}_cam_{f}"] = numpy.asarray(v, dtype=numpy.double)
            out[key + "_origins"] = o
            out[key + "_dirs"] = d
    save("camera", seed=numpy.array(seed), names=numpy.array(names), **out)


def probe_rays(ref, camera, object_groups, width, height, seed):
    """Ray batch for intersection fixtures: primary camera rays, the diffuse
    bounce rays leaving their hit points (origins ON surfaces: exercises
    t_min and rays starting inside/under geometry), and rays between random
    points of the scene's bounding box."""
    saved = ref.camera.RNG
    ref.camera.RNG = KeyedCameraRNG(seed, 0)
    try:
        o, d = camera.get_ray_components(width, height, 1)
    finally:
        ref.camera.RNG = saved
    o = o.reshape(-1, 3)
    d = d.reshape(-1, 3)
    with quiet():
        hit, t, p, n, uv, mat, back, grp = merge_hits(object_groups, o, d)
    px, py, ps = path_identity(width, height, 1)
    vec = keyed_rng.keyed_unit_vectors(seed, px[hit], py[hit], ps[hit], 0)
    flip = numpy.einsum("ij,ij->i", vec, n[hit]) < 0.0
    vec[flip] *= -1.0
    o2 = p[hit].astype(numpy.single)
    d2 = vec
    # reflected rays too (coherent secondary rays)
    dn = numpy.einsum("ij,ij->i", d[hit], n[hit])
    d3 = (d[hit] - 2.0 * dn[:, None] * n[hit]).astype(numpy.single)
    rng = numpy.random.default_rng(seed)
    pts = p[hit]
    lo, hi = pts.min(axis=0) - 0.5, pts.max(axis=0) + 0.5
    lo = numpy.maximum(lo, -30.0)
    hi = numpy.minimum(hi, 30.0)
    m = 400
    a = rng.uniform(lo, hi, size=(m, 3))
    b = rng.uniform(lo, hi, size=(m, 3))
    d4 = b - a
    d4 /= numpy.linalg.norm(d4, axis=1)[:, None]
    origins = numpy.concatenate([o, o2, o2, a.astype(numpy.single)]).astype(numpy.single)
    dirs = numpy.concatenate([d, d2, d3, d4.astype(numpy.single)]).astype(numpy.single)
    return origins, dirs


def write_intersect(ref, name, scene_fn, width=40, height=22, seed=99, texture_names=()):
    aspect = width / height
    with quiet():
        camera, groups, material_map = scene_fn(aspect)
    o, d = probe_rays(ref, camera, groups, width, height, seed)
    out = {}
    spy_s, spy_t = NumpyProxy(), NumpyProxy()
    saved = (ref.sphere_group.numpy, ref.triangle_group.numpy)
    for gi, g in enumerate(groups):
        with quiet():
            res = g.get_hits(o, d, 0.001, 1000.0)
        for f, v in zip(("hit", "t", "p", "n", "uv", "mat", "back"), res):
            out[f"g{gi}_{f}"] = numpy.asarray(v)
        # primitive index: argmin inside _get_hits on ALL rays (cull and
        # chunking do not change results: triangle_group.py:158-166,248-254)
        if hasattr(g, "_get_hits"):
            ref.sphere_group.numpy, ref.triangle_group.numpy = spy_s, spy_t
            try:
                with quiet():
                    raw = g._get_hits(o, d, 0.001, 1000.0)
            finally:
                ref.sphere_group.numpy, ref.triangle_group.numpy = saved
            spy = spy_s if isinstance(g, ref.sphere_group.SphereGroup) else spy_t
            prim = numpy.where(raw[0], spy.last_argmin, -1).astype(numpy.int32)
            # with the cull a hit can only disappear, never change
            assert numpy.all(~res[0] | raw[0])
            out[f"g{gi}_prim"] = numpy.where(res[0], prim, -1).astype(numpy.int32)
    with quiet():
        merged = merge_hits(groups, o, d)
    for f, v in zip(("hit", "t", "p", "n", "uv", "mat", "back", "group"), merged):
        out[f"merged_{f}"] = v
    prim = numpy.full(o.shape[0], -1, dtype=numpy.int32)
    for gi in range(len(groups)):
        if f"g{gi}_prim" in out:
            sel = merged[7] == gi
            prim[sel] = out[f"g{gi}_prim"][sel]
        else:
            prim[merged[7] == gi] = 0
    out["merged_prim"] = prim
    flat = flatten_scene(ref, camera, groups, material_map)
    files = texture_file_map(ref, material_map, texture_names)
    save("intersect_" + name, origins=o, dirs=d, t_min=numpy.array(0.001), t_max=numpy.array(1000.0),
         **scene_io.pack_scene(flat, texture_files=files), **out)


def write_scatter(ref):
    """Every material's scatter on injected hit records, with the dtypes
    main.render() hands over (float32 directions; float64 arrays holding
    float32 values for points, normals and uvs: main.py:117-120,189-195)."""
    M = ref.materials
    rng = numpy.random.default_rng(31337)
    n = 600
    seed = 777

    def unit(v):
        return v / numpy.linalg.norm(v, axis=1)[:, None]

    normals = unit(rng.normal(size=(n, 3))).astype(numpy.single)
    dirs = unit(rng.normal(size=(n, 3))).astype(numpy.single)
    # hit normals always face the incoming ray
    facing = numpy.einsum("ij,ij->i", dirs, normals) > 0
    normals[facing] *= -1.0
    # a block of grazing rays (total internal reflection from inside glass)
    tangent = unit(numpy.cross(normals[:150], rng.normal(size=(150, 3))))
    graze = unit(tangent - 0.05 * rng.uniform(0.0, 4.0, size=(150, 1)) * normals[:150])
    dirs[:150] = graze.astype(numpy.single)
    # a block of axis-aligned normals (discrete normal colours, 0.8 threshold)
    axes = numpy.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=numpy.single)
    tilt = unit(axes[numpy.arange(120) % 6] + rng.uniform(-0.9, 0.9, size=(120, 3))).astype(numpy.single)
    normals[150:270] = tilt
    facing = numpy.einsum("ij,ij->i", dirs, normals) > 0
    dirs[facing] *= -1.0
    points = rng.uniform(-6.0, 6.0, size=(n, 3)).astype(numpy.single)
    points[:40] = numpy.round(points[:40] * 2.0) / 2.0  # on checker cell borders
    uvs = rng.uniform(-0.2, 1.2, size=(n, 2)).astype(numpy.single)
    uvs[:64, 0] = (numpy.arange(64) / 63.0)
    uvs[:64, 1] = (numpy.arange(64)[::-1] / 63.0)
    uvs[64:80] = [[0.0, 0.0]]
    uvs[80:96] = [[1.0, 1.0]]
    back = rng.uniform(size=n) < 0.5
    back[:150] = True
    px = rng.integers(0, 1920, size=n)
    py = rng.integers(0, 1080, size=n)
    ps = rng.integers(0, 64, size=n)

    tex = M.TexturedDiffuse("uv_test.jpg")
    mats = {
        "diffuse": M.Diffuse(numpy.array([0.2, 0.7, 0.1], dtype=numpy.single)),
        "textured": tex,
        "checker": M.CheckerboardDiffuse(numpy.array([2.0, 2.0, 2.0]), numpy.array([0.2, 0.0, 0.1]),
                                         numpy.array([0.5, 0.5, 0.5]), numpy.array([0.8, 0.8, 0.8])),
        "normal_rgb": M.NormalToRGBDiffuse(),
        "normal_discrete": M.NormalToDiscreteRGBDiffuse(),
        "metal": M.Metal(numpy.array([0.8, 0.8, 0.8], dtype=numpy.single), 0.0),
        "metal_fuzzy": M.Metal(numpy.array([0.7, 0.6, 0.5], dtype=numpy.single), 0.35),
        "dielectric": M.Dielectric(1.5),
        "dielectric_water": M.Dielectric(1.33),
    }
    out = dict(seed=numpy.array(seed), dirs=dirs, points=points.astype(numpy.double),
               normals=normals.astype(numpy.double), uvs=uvs.astype(numpy.double), back=back,
               px=px, py=py, ps=ps, names=numpy.array(list(mats)))
    flat_mats = {}
    ctx = ScatterContext(seed)
    saved = (M.RNG, M.numpy_random_unit_vecs)
    M.RNG, M.numpy_random_unit_vecs = ctx, ctx.unit_vecs
    try:
        for k, (name, mat) in enumerate(mats.items()):
            bounce = k % 5
            ctx.set(px, py, ps, bounce)
            o, d, c, a = mat.scatter(dirs, points.astype(numpy.double), normals.astype(numpy.double),
                                     uvs.astype(numpy.double), back)
            out[name + "_bounce"] = numpy.array(bounce)
            out[name + "_origins"] = numpy.asarray(o)
            out[name + "_dirs64"] = numpy.asarray(d, dtype=numpy.double)
            # what main.py:198-199 stores: float32
            out[name + "_dirs"] = numpy.asarray(d).astype(numpy.single)
            out[name + "_colours"] = numpy.asarray(c).astype(numpy.single)
            out[name + "_absorbed"] = numpy.asarray(a)
            flat_mats[k] = flatten_scene(ref, types.SimpleNamespace(
                camera_pos=0, U=0, V=0, W=0, viewport_horizontal=0, viewport_vertical=0,
                bottomleft_focalplane_pos=0, lens_radius=0), [], {k: mat})["materials"][k]
    finally:
        M.RNG, M.numpy_random_unit_vecs = saved
    scene = dict(camera=dict(pos=numpy.zeros(3), U=numpy.zeros(3), V=numpy.zeros(3), W=numpy.zeros(3),
                             horiz=numpy.zeros(3), vert=numpy.zeros(3), bottom_left=numpy.zeros(3),
                             lens_radius=0.0),
                 groups=[], materials=flat_mats,
                 sky=dict(horizon=ref.main.HORIZON_COLOUR, zenith=ref.main.SKY_COLOUR))
    save("scatter", **scene_io.pack_scene(scene, texture_files={1: "uv_test.jpg"}), **out)


def write_frame(ref, name, scene_fn, width, height, samples, bounces, seed, texture_names=(), s_begin=0,
                with_trace=False):
    aspect = width / height
    with quiet():
        camera, groups, material_map = scene_fn(aspect)
        trace = [] if with_trace else None
        res = harness_render(ref, camera, groups, material_map, width, height, samples, bounces,
                             seed=seed, s_begin=s_begin, trace=trace)
    flat = flatten_scene(ref, camera, groups, material_map)
    files = texture_file_map(ref, material_map, texture_names)
    extra = {}
    if with_trace:
        for rec in trace:
            b = rec["bounce"]
            for f, v in rec.items():
                if f != "bounce":
                    extra[f"trace_b{b}_{f}"] = v
    save("frame_" + name, size=numpy.array([width, height, samples, bounces, s_begin]), seed=numpy.array(seed),
         image=res["image"], path_colours=res["path_colours"], counts=res["counts"], segments=res["segments"],
         sample_sum=res["sample_sum"], **scene_io.pack_scene(flat, texture_files=files), **extra)
    print(f"   {name}: per-bounce active counts {res['counts'].tolist()}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma list of: check,kat,camera,intersect,scatter,frames,heavy,composite")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))

    def want(k):
        return not only or k in only

    os.makedirs(GOLDEN, exist_ok=True)
    ref = import_reference()
    S = ref.scenes
    records = sphere_field_records()
    if want("check"):
        self_check(ref)
    if want("kat"):
        write_kat(ref)
    if want("camera"):
        write_camera(ref)
    if want("scatter"):
        write_scatter(ref)
    if want("intersect"):
        write_intersect(ref, "demo", S.numpy_one_weekend_demo_scene)
        write_intersect(ref, "cover187", lambda a: cover_scene(ref, a, records))
        write_intersect(ref, "glass_experiment", S.numpy_glass_experiment_scene)
        write_intersect(ref, "triangles", S.numpy_triangles_scene)
        write_intersect(ref, "bunny", S.ray_group_triangle_group_bunny_scene, texture_names=["bunnyTexture.tif"])
        write_intersect(ref, "icospheres", S.sphere_types_test_scene)
        write_intersect(ref, "cloth_bunny", lambda a: cloth_bunny_scene(ref, a),
                        texture_names=["bunnyTexture.tif", "picnicCloth.jpg"])
        write_intersect(ref, "disks", S.disk_test_scene, texture_names=["uv_test.jpg"])
    if want("frames"):
        write_frame(ref, "demo", S.numpy_one_weekend_demo_scene, 64, 36, 4, 5, seed=1234, with_trace=True)
        write_frame(ref, "demo_shard", S.numpy_one_weekend_demo_scene, 64, 36, 2, 5, seed=1234, s_begin=2)
        write_frame(ref, "cover187", lambda a: cover_scene(ref, a, records), 64, 36, 4, 10, seed=2024)
        write_frame(ref, "glass_experiment", S.numpy_glass_experiment_scene, 64, 36, 4, 10, seed=5)
        write_frame(ref, "simple_sphere", S.numpy_simple_sphere_scene, 48, 27, 4, 5, seed=6)
        write_frame(ref, "triangles", S.numpy_triangles_scene, 48, 27, 4, 5, seed=7)
        write_frame(ref, "bunny_textured", S.ray_group_triangle_group_bunny_scene, 64, 36, 4, 10, seed=8,
                    texture_names=["bunnyTexture.tif"], with_trace=True)
        write_frame(ref, "bunny_diffuse", lambda a: bunny_diffuse_scene(ref, a), 64, 36, 4, 10, seed=9,
                    texture_names=[])
        write_frame(ref, "cloth_bunny", lambda a: cloth_bunny_scene(ref, a), 64, 36, 4, 10, seed=10,
                    texture_names=["bunnyTexture.tif", "picnicCloth.jpg"])
        write_frame(ref, "texture_test", S.texture_test_scene, 48, 27, 4, 5, seed=11, texture_names=["uv_test.jpg"])
        write_frame(ref, "smooth_normal", S.smooth_normal_test_scene, 48, 27, 4, 5, seed=12)
        write_frame(ref, "icospheres", S.sphere_types_test_scene, 48, 27, 4, 5, seed=13)
        write_frame(ref, "disks", S.disk_test_scene, 48, 27, 4, 5, seed=14, texture_names=["uv_test.jpg"])
    if want("heavy"):
        write_intersect(ref, "bunnies", S.numpy_bunnies_scene, width=32, height=18)
        write_intersect(ref, "cow", S.numpy_cow_scene, width=32, height=18)
        write_frame(ref, "bunnies", S.numpy_bunnies_scene, 48, 27, 2, 6, seed=15)
        write_frame(ref, "cow", S.numpy_cow_scene, 32, 18, 2, 4, seed=16)
    if want("composite"):
        write_intersect(ref, "cow_teapot", lambda a: cow_teapot_field_scene(ref, a, records), width=32, height=18)
        write_frame(ref, "cow_teapot", lambda a: cow_teapot_field_scene(ref, a, records), 40, 22, 1, 4, seed=17)
        write_smooth_normals(ref, "teapot.obj", 18)
        write_smooth_normals(ref, "cow.obj", 23)


if __name__ == "__main__":
    main()
