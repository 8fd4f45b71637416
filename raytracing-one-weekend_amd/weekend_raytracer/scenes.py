"""
Scene functions: f(aspect_ratio) -> (camera, [object groups], {index: material}).

The first block rebuilds the reference's scenes (reference scenes.py) with the
same numbers, dtypes and group order, so the float32 arrays handed to the GPU
are the ones the reference would build.  The second block holds the composite
scenes of the benchmark configurations, which the reference does not ship as
functions.

Asset files are looked up in the current directory first (the reference
opens bare file names, scenes.py:57,236,588,613) and then in ASSET_DIR.
"""
import json
import os
import random

import numpy

from . import materials
from .camera import Camera
from .disk import Disk
from .obj_tri_mesh import OBJTriMesh
from .sphere_group import SphereGroup
from .triangle_group import TriangleGroup

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")

f32 = numpy.single


def asset(name):
    return name if os.path.exists(name) else os.path.join(ASSET_DIR, name)


def _v(*xyz, dtype=None):
    return numpy.array(xyz, dtype=dtype)


def _camera(pos, lookat, focus, aperture, aspect_ratio, hfov, dtype=None):
    return Camera(numpy.array(pos, dtype=dtype), numpy.array(lookat, dtype=dtype), focus, aperture, aspect_ratio,
                  hfov)


def _spheres(*rows):
    """rows: (centre, radius, stored colour, material index)."""
    grp = SphereGroup()
    for centre, radius, colour, material in rows:
        grp.add_sphere(numpy.array(centre, dtype=f32), radius, numpy.array(colour, dtype=f32), material)
    return grp


GROUND_1000 = ((0.0, -1000.0, 0.0), 1000.0)


def _checker(scale, offset, colour_a, colour_b, dtype=None):
    return materials.CheckerboardDiffuse(numpy.array([scale] * 3, dtype=dtype), numpy.array([offset] * 3, dtype=dtype),
                                         numpy.array(colour_a, dtype=dtype), numpy.array(colour_b, dtype=dtype))


def _diffuse(*rgb):
    return materials.Diffuse(numpy.array(rgb, dtype=f32))


def _metal(value, fuzz=0.0):
    return materials.Metal(numpy.array([value] * 3, dtype=f32), fuzz)


def _read(obj_name):
    mesh = OBJTriMesh()
    mesh.read(asset(obj_name))
    return mesh


def mesh_group(mesh, material_id, offset=(0.0, 0.0, 0.0), lift=True, point_dtype=None, uvs=False, normals=None):
    """Fill a TriangleGroup from a mesh.  Vertex coordinates are shifted as
    Python floats (lift: minus the mesh's lowest y) before the point array is
    made in point_dtype (None = float64).  normals: None (default +Y),
    "smooth" (OBJTriMesh.get_smooth_vertex_normal) or "file" (the vn records)."""
    drop = min(v[1] for v in mesh.vertices) if lift else None
    grp = TriangleGroup(material_id)
    for tri in mesh.faces:
        pts = []
        for k in range(3):
            x, y, z = mesh.vertices[tri[k][0]]
            if lift:
                pts.append(numpy.array([x + offset[0], y - drop, z + offset[2]], dtype=point_dtype)
                           if any(offset) else numpy.array([x, y - drop, z], dtype=point_dtype))
            else:
                pts.append(numpy.array(mesh.vertices[tri[k][0]], dtype=point_dtype))
        extra = {}
        if uvs:
            for k in range(3):
                extra[f"uv{k}"] = numpy.array(mesh.uvs[tri[k][1]])
        if normals == "smooth":
            for k in range(3):
                extra[f"normal{k}"] = mesh.get_smooth_vertex_normal(tri[k][0])
        elif normals == "file":
            for k in range(3):
                extra[f"normal{k}"] = numpy.array(mesh.vertex_normals[tri[k][2]])
        grp.add_triangle(*pts, **extra)
    return grp


# --------------------------------------------------------------------------
# the reference's scenes
# --------------------------------------------------------------------------
def numpy_bunnies_scene(aspect_ratio):
    """Four bunnies: metal, glass, red/blue checker, normal-coloured."""
    camera = _camera([3.0, 5.0, 10.0], [-1.0, 1.2, 0.0], 10, 0.0, aspect_ratio, 60.0)
    material_map = {
        0: _checker(1.0, 0.0, [0.5, 0.5, 0.5], [0.3, 0.3, 0.3]),
        1: _checker(2.0, 0.2, [0.7, 0.3, 0.2], [0.1, 0.2, 0.5]),
        2: materials.Metal(numpy.array([0.8, 0.8, 0.8]), 0.0),
        3: materials.Dielectric(1.5),
        4: materials.NormalToRGBDiffuse(),
    }
    ground = _spheres((*GROUND_1000, (0, 0, 0), 0))
    mesh = _read("bunny.obj")
    spacing = 2.0
    placed = [(2, -spacing, -spacing), (3, spacing, spacing), (1, -spacing, spacing), (4, spacing, -spacing)]
    bunnies = [mesh_group(mesh, mat, offset=(ox, 0.0, oz), point_dtype=f32) for mat, ox, oz in placed]
    return camera, [ground] + bunnies, material_map


def numpy_cow_scene(aspect_ratio):
    camera = _camera([11.0, 8.0, 9.0], [1.0, 3.0, -1.0], 10, 0.0, aspect_ratio, 60.0)
    material_map = {
        0: _checker(1.0, 0.0, [0.2, 0.7, 0.3], [0.1, 0.9, 0.2]),
        1: _checker(1.0, 0.2, [0.8, 0.8, 0.85], [0.25, 0.25, 0.2]),
    }
    ground = _spheres((*GROUND_1000, (0, 0, 0), 0))
    cow = mesh_group(_read("cow.obj"), 1, point_dtype=f32)
    return camera, [ground, cow], material_map


def numpy_one_weekend_demo_scene(aspect_ratio):
    """Three unit spheres (normal-coloured, glass, mirror) on a grey ground."""
    camera = _camera([13.0, 2.0, 3.0], [0.0, 0.5, 0.0], 10.0, 0.0, aspect_ratio, 30.0)
    material_map = {
        0: _diffuse(0.5, 0.5, 0.5),
        1: _metal(0.9),
        2: materials.Dielectric(1.5),
        3: materials.NormalToDiscreteRGBDiffuse(),
    }
    spheres = _spheres(
        ((-4, 1, 0), 1.0, (1, 0, 0), 3),
        ((0, 1, 0), 1.0, (0, 1, 0), 2),
        ((4, 1, 0), 1.0, (0.9, 0.9, 0.9), 1),
        ((0, -1000, 0), 1000.0, (0.5, 0.5, 0.5), 0),
    )
    return camera, [spheres], material_map


def numpy_glass_experiment_scene(aspect_ratio):
    camera = _camera([0, 0, 0], [0.0, 0.0, -5.0], 10.0, 0.0, aspect_ratio, 90.0, dtype=f32)
    material_map = {
        0: _diffuse((148 / 256), (116 / 256), (105 / 256)),
        1: _diffuse(0.1, 0.2, 0.5),
        2: materials.NormalToDiscreteRGBDiffuse(),
        3: _metal(0.8),
        4: materials.Dielectric(1.5),
    }
    grey = (0.5, 0.5, 0.5)
    rows = [
        ((-6.0, 0.0, -10.0), 3.0, grey, 4),       # glass
        ((0.0, 0.0, -10.0), 3.0, grey, 1),        # blue
        ((6.0, 0.0, -10.0), 3.0, grey, 2),        # normal-coloured
        ((5.0, 6.0, -16.0), 3.0, grey, 3),        # floating mirror
        ((-9.0, -3.0, -16.0), 3.0, grey, 2),      # half buried behind the glass
    ]
    rows += [(((x * 1.3) - 12.0, (y * 2.0) + 1.5, -16.0), 0.3, grey, 2) for x in range(3) for y in range(3)]
    rows.append(((0.0, -503.0, -10.0), 500, grey, 0))
    return camera, [_spheres(*rows)], material_map


def numpy_triangles_scene(aspect_ratio):
    camera = _camera([0.0, 1.0, 6.0], [0.0, 0.5, 0.0], 10, 0.0, aspect_ratio, 50.0)
    material_map = {0: _diffuse(0.5, 0.5, 0.5)}
    grp = TriangleGroup(0)
    grp.add_triangle(_v(1.0, 0.0, 0.0, dtype=f32), _v(2.0, 0.0, 0.0, dtype=f32), _v(1.0, 2.0, 0.0, dtype=f32))
    grp.add_triangle(_v(-2.5, 0.0, 0.0, dtype=f32), _v(-1.5, 0.0, 0.0, dtype=f32), _v(-2.0, 0.75, 0.0, dtype=f32))
    back = _v(0.0, 0.0, -2.0, dtype=f32)
    grp.add_triangle(_v(-2.0, 0.0, -1.0, dtype=f32) + back, _v(2.0, 0.0, 1.0, dtype=f32) + back,
                     _v(0.0, 2.0, 0.0, dtype=f32) + back)
    grp.add_triangle(_v(-200, 0, 200, dtype=f32), _v(200, 0, 200, dtype=f32), _v(0, 0, -200, dtype=f32))
    return camera, [grp], material_map


def numpy_simple_sphere_scene(aspect_ratio):
    camera = _camera([10.0, 5.0, 10.0], [0.0, 0.5, 0.0], 10, 0.0, aspect_ratio, 50.0)
    material_map = {
        0: _diffuse(0.5, 0.5, 0.5),
        1: _checker(2, 0, [0.5, 0.8, 0.5], [0.9, 0.5, 0.5], dtype=f32),
    }
    red = (1, 0, 0)
    spheres = _spheres((*GROUND_1000, red, 0), ((-1.0, 0.5, 0.0), 0.5, red, 0), ((2, 2, 0), 2.0, red, 1))
    return camera, [spheres], material_map


def _bunny_materials():
    return {
        0: _checker(2.0, 0.0, [0.5, 0.5, 0.5], [0.8, 0.8, 0.8]),
        1: materials.TexturedDiffuse(asset("bunnyTexture.tif")),
        2: _metal(0.8),
    }


def ray_group_triangle_group_bunny_scene(aspect_ratio):
    """Textured bunny with smooth vertex normals on a checkerboard ground."""
    camera = _camera([-2.0, 3.5, 8.0], [-2.2, 1.7, 0.0], 10, 0.0, aspect_ratio, 53.0)
    bunny = mesh_group(_read("bunny.obj"), 1, uvs=True, normals="smooth")
    ground = _spheres((*GROUND_1000, (1, 0, 0), 0))
    return camera, [bunny, ground], _bunny_materials()


def texture_test_scene(aspect_ratio):
    camera = _camera([2.5, 2.5, 2.5], [0.5, 0.0, 0.5], 10, 0.0, aspect_ratio, 50.0)
    material_map = {0: _diffuse(0.5, 0.5, 0.5), 1: materials.TexturedDiffuse(asset("uv_test.jpg"))}
    square = mesh_group(_read("square.obj"), 1, uvs=True)
    ground = _spheres((*GROUND_1000, (1, 0, 0), 0))
    return camera, [square, ground], material_map


def smooth_normal_test_scene(aspect_ratio):
    camera = _camera([2.5, 2.5, 2.5], [0.5, 0.5, 0.5], 10, 0.0, aspect_ratio, 50.0)
    material_map = {0: _checker(4.0, 0.0, [0.5, 0.5, 0.5], [0.8, 0.8, 0.8]), 1: _metal(0.9)}
    tris = mesh_group(_read("angled_tris_standing.obj"), 1, uvs=True, normals="smooth")
    ground = _spheres((*GROUND_1000, (1, 0, 0), 0))
    return camera, [tris, ground], material_map


def _test_materials():
    return {
        0: _diffuse(0.5, 0.5, 0.5),
        1: _checker(4.0, 0.0, [1.0, 0.3, 0.3], [0.2, 1.0, 0.3]),
        2: _metal(0.9),
        3: materials.NormalToRGBDiffuse(),
    }


def blender_cylinder_vert_normals_test_scene(aspect_ratio):
    """As shipped the reference returns only the ground sphere of this scene
    (scenes.py:908-909); the cylinder groups are available through
    blender_cylinder_groups()."""
    camera = _camera([1.5, 1.5, 1.5], [-0.25, 0.0, -0.25], 10, 0.0, aspect_ratio, 35.0)
    ground = _spheres((*GROUND_1000, (1, 0, 0), 0))
    return camera, [ground], _test_materials()


def blender_cylinder_groups():
    return [mesh_group(_read(name), 2, lift=False, uvs=True, normals="file")
            for name in ("cylinder_smooth.obj", "cylinder_faceted.obj")]


def sphere_types_test_scene(aspect_ratio):
    """Smooth and faceted icospheres (normals from the file) around a true sphere."""
    camera = _camera([0, 3, 7], [0.0, 1, 0.0], 10, 0.0, aspect_ratio, 50.0)
    smooth = mesh_group(_read("smooth-icosphere.obj"), 2, lift=False, uvs=True, normals="file")
    faceted = mesh_group(_read("faceted-icosphere.obj"), 2, lift=False, uvs=True, normals="file")
    spheres = _spheres((*GROUND_1000, (1, 0, 0), 1), ((0, 1.0, 0), 1.0, (1, 0, 0), 2))
    return camera, [smooth, faceted, spheres], _test_materials()


def disk_test_scene(aspect_ratio):
    camera = _camera([3.0, 3.0, 3.0], [0.0, 0.0, 0.0], 10, 0.0, aspect_ratio, 40.0)
    material_map = {
        0: _diffuse(0.5, 0.5, 0.5),
        1: _diffuse(0.3, 0.8, 0.1),
        2: _diffuse(0.8, 0.15, 0.2),
        3: materials.TexturedDiffuse(asset("uv_test.jpg")),
    }
    ground = _spheres((*GROUND_1000, (1, 0, 0), 0))

    def unit32(*xyz):
        n = numpy.array(xyz, dtype=f32)
        return (n / numpy.linalg.norm(n)).astype(f32)

    upright = Disk(numpy.array([0, 0.5, 0], dtype=f32), 0.5, unit32(1, 0, 0), 3)
    flat = Disk(numpy.array([0, 0.15, -1.5], dtype=f32), 1.5, unit32(0, 1, 0), 3,
                up=numpy.array([0, 0, -1], dtype=f32))
    return camera, [ground, upright, flat], material_map


# --------------------------------------------------------------------------
# benchmark configurations (composites of the reference's pieces)
# --------------------------------------------------------------------------
def sphere_field_records(count=None, seed=20211):
    """count=None: the 187 records of the shipped sphere_data.json.  Otherwise
    a field generated with the recipe of the reference's generator
    (dev_helpers/sandbox.py:40-97) on a grid wide enough for `count` spheres
    after the rejections around the three big spheres, from a fixed seed."""
    if count is None:
        with open(asset("sphere_data.json")) as handle:
            return json.load(handle)
    rnd = random.Random(seed)
    half = 11
    while True:
        records = []
        for x in range(-half, half):
            for z in range(-half, half):
                x_pos = x + (rnd.random() * 0.7)
                z_pos = z + (rnd.random() * 0.7)
                if any((bx - 1.1 < x_pos < bx + 1.1) and (-1.1 < z_pos < 1.1) for bx in (-4, 0, 4)):
                    continue
                radius = rnd.uniform(0.17, 0.23)
                rec = dict(pos=[x_pos, radius, z_pos], radius=radius)
                choice = rnd.random()
                if choice < 0.8:
                    rec.update(material="diffuse", colour=[rnd.random() * rnd.random() for _ in range(3)])
                elif choice < 0.95:
                    rec.update(material="metal", colour=[rnd.uniform(0.5, 1.0) for _ in range(3)],
                               fuzziness=rnd.uniform(0, 0.5))
                else:
                    rec.update(material="glass", ior=1.5)
                records.append(rec)
        if len(records) >= count:
            return records[:count]
        half += 1


def cover_scene(aspect_ratio, records=None, aperture=0.1, palette=250):
    """Book-1 cover: ground + lambertian / glass / metal unit spheres + a field
    of small spheres under the depth-of-field camera of the demo scene
    (reference scenes.py:276-283 with the aperture of its line 280).  Every
    small sphere gets its own material while they fit the reference's uint8
    material indices; beyond `palette` materials the diffuse and metal
    colours (and fuzziness) are shared round-robin."""
    records = sphere_field_records() if records is None else records
    camera = _camera([13.0, 2.0, 3.0], [0.0, 0.5, 0.0], 10.0, aperture, aspect_ratio, 30.0)
    material_map = {
        0: _diffuse(0.5, 0.5, 0.5),
        1: _diffuse(0.4, 0.2, 0.1),
        2: materials.Dielectric(1.5),
        3: materials.Metal(numpy.array([0.7, 0.6, 0.5], dtype=f32), 0.0),
    }
    black = (0, 0, 0)
    rows = [((0, -1000, 0), 1000.0, black, 0), ((-4, 1, 0), 1.0, black, 1), ((0, 1, 0), 1.0, black, 2),
            ((4, 1, 0), 1.0, black, 3)]
    first_small = 4
    for i, rec in enumerate(records):
        index = first_small + (i % palette)
        if index not in material_map:
            if rec["material"] == "diffuse":
                material_map[index] = materials.Diffuse(numpy.array(rec["colour"], dtype=f32))
            elif rec["material"] == "metal":
                material_map[index] = materials.Metal(numpy.array(rec["colour"], dtype=f32), rec["fuzziness"])
            else:
                material_map[index] = materials.Dielectric(rec["ior"])
        rows.append((rec["pos"], rec["radius"], black, index))
    return camera, [_spheres(*rows)], material_map


def bunny_diffuse_scene(aspect_ratio):
    """The bunny scene with the Lambertian bunny material the reference
    defines but leaves unused (scenes.py:584-586) instead of the texture."""
    camera, groups, material_map = ray_group_triangle_group_bunny_scene(aspect_ratio)
    material_map[1] = _diffuse(0.2, 0.7, 0.1)
    return camera, groups, material_map


def cloth_bunny_scene(aspect_ratio):
    """Textured bunny plus the UV-mapped picnic cloth mesh."""
    camera, groups, material_map = ray_group_triangle_group_bunny_scene(aspect_ratio)
    material_map[3] = materials.TexturedDiffuse(asset("picnicCloth.jpg"))
    cloth = mesh_group(_read("smallCloth.obj"), 3, lift=False, uvs=True)
    return camera, [groups[0], cloth, groups[1]], material_map


def cow_teapot_field_scene(aspect_ratio, field=187):
    """Cow and teapot meshes over the small-sphere field (no reference scene
    uses teapot.obj; materials follow the cow scene and the cover scene)."""
    camera = _camera([13.0, 6.0, 9.0], [0.0, 1.5, 0.0], 10.0, 0.0, aspect_ratio, 40.0)
    _, (spheres,), material_map = cover_scene(aspect_ratio, sphere_field_records(None if field == 187 else field),
                                              palette=240)
    material_map[250] = _checker(1.0, 0.2, [0.8, 0.8, 0.85], [0.25, 0.25, 0.2])
    material_map[251] = _metal(0.85)
    cow = mesh_group(_read("cow.obj"), 250, offset=(-3.0, 0.0, -2.5), point_dtype=f32)
    teapot = mesh_group(_read("teapot.obj"), 251, offset=(3.0, 0.0, 3.5), point_dtype=f32)
    return camera, [spheres, cow, teapot], material_map
