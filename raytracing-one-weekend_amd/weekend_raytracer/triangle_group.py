"""
TriangleGroup: SoA triangle builder with per-vertex normals/uvs and a
bounding sphere (reference triangle_group.py:31-119); get_hits runs the
bounding-sphere cull + Moller-Trumbore closest-hit kernel with smooth-normal
and uv interpolation (reference triangle_group.py:150-452).

Building is linear: triangles are collected in lists and the float32 arrays
and the bounding sphere are produced once, by the same numpy operations the
reference applies after every single add_triangle.
"""
import numpy

from . import _native

_FIELDS = ("pt0s", "pt1s", "pt2s", "uv0s", "uv1s", "uv2s", "normal0s", "normal1s", "normal2s", "As", "Bs",
           "normals")


class TriangleGroup():
    def __init__(self, material_id):
        self.material_id = material_id
        self._rows = {f: [] for f in _FIELDS}
        self._cache = None
        self._bounds = None

    def add_triangle(self, pt0, pt1, pt2, uv0=(0.0, 0.0), uv1=(0.0, 0.0), uv2=(0.0, 0.0),
                     normal0=(0, 1, 0), normal1=(0, 1, 0), normal2=(0, 1, 0)):
        """Points are counter-clockwise seen from the side the face normal
        points to.  Edges and the face normal are formed in the caller's
        dtype and only then stored as float32, as the reference does."""
        A = pt1 - pt0
        B = pt2 - pt0
        face = numpy.cross(A, B)
        face /= numpy.linalg.norm(face)
        for f, v in zip(_FIELDS, (pt0, pt1, pt2, uv0, uv1, uv2, normal0, normal1, normal2, A, B, face)):
            self._rows[f].append(numpy.array(v, dtype=numpy.single))
        self._cache = None
        self._bounds = None

    @property
    def num_triangles(self):
        return len(self._rows["pt0s"])

    def _arrays(self):
        if self._cache is None:
            self._cache = {f: numpy.array(rows, dtype=numpy.single) for f, rows in self._rows.items()}
        return self._cache

    def __getattr__(self, name):
        # pt0s ... normals: float32 arrays, None while the group is empty
        if name in _FIELDS:
            return self._arrays()[name] if self.num_triangles else None
        raise AttributeError(name)

    def update_sphere_bounds(self):
        """Centre = mean of all vertices, radius = 1.01 x the largest
        distance to it (reference triangle_group.py:107-119)."""
        a = self._arrays()
        all_pts = numpy.concatenate((a["pt0s"], a["pt1s"], a["pt2s"]), axis=0)
        centre = numpy.mean(all_pts, axis=0)
        rel = all_pts - centre
        self._bounds = (centre, numpy.max(numpy.sqrt(numpy.einsum("ij,ij->i", rel, rel))) * 1.01)

    @property
    def bounds_centre(self):
        if self._bounds is None and self.num_triangles:
            self.update_sphere_bounds()
        return self._bounds[0] if self._bounds else None

    @property
    def bounds_radius(self):
        if self._bounds is None and self.num_triangles:
            self.update_sphere_bounds()
        return self._bounds[1] if self._bounds else None

    def _flat(self):
        a = self._arrays()
        return dict(type="triangles", pt0=a["pt0s"], A=a["As"], B=a["Bs"], n0=a["normal0s"], n1=a["normal1s"],
                    n2=a["normal2s"], uv0=a["uv0s"], uv1=a["uv1s"], uv2=a["uv2s"],
                    material_id=int(self.material_id), bounds_centre=self.bounds_centre,
                    bounds_radius=self.bounds_radius)

    def get_hits(self, ray_origins, ray_dirs, t_min, t_max):
        """-> (ray_hits, hit_ts, hit_pts, hit_normals, hit_uvs, hit_material_indecies, back_facing).
        Rays without a hit carry zeros and t = t_max + 1; hit_ts is float64
        holding float32 values, like the reference's (triangle_group.py:172)."""
        ctx = _native.resident_scene("groups", lambda: dict(groups=[self._flat()]))
        r = ctx.intersect(0, ray_origins, ray_dirs, t_min, t_max)
        return r["hit"], r["t"].astype(numpy.double), r["p"], r["n"], r["uv"], r["mat"], r["back"]
