"""
SphereGroup: SoA sphere builder (reference sphere_group.py:14-45) whose
get_hits runs the ray x sphere closest-hit kernel (reference
sphere_group.py:72-261).  No dense (rays x spheres) grid is materialised, so
the reference's host-memory chunking (sphere_group.py:47-120) has no
counterpart.
"""
import numpy

from . import _native


class SphereGroup():
    def __init__(self):
        self._centres, self._radii, self._colours, self._materials = [], [], [], []
        self._cache = None

    def add_sphere(self, centre, radius, colour, material_index):
        self._centres.append(numpy.array(centre, dtype=numpy.single))
        self._radii.append(numpy.single(radius))
        self._colours.append(numpy.array(colour, dtype=numpy.single))
        self._materials.append(numpy.array(material_index).astype(numpy.ubyte))
        self._cache = None

    def _arrays(self):
        if self._cache is None and self._radii:
            self._cache = (numpy.array(self._centres, dtype=numpy.single).reshape(-1, 3),
                           numpy.array(self._radii, dtype=numpy.single),
                           numpy.array(self._colours, dtype=numpy.single).reshape(-1, 3),
                           numpy.array(self._materials, dtype=numpy.ubyte))
        return self._cache

    # the reference's attributes (None while empty, sphere_group.py:19-23)
    @property
    def num_spheres(self):
        return len(self._radii)

    @property
    def centres(self):
        return self._arrays()[0] if self._radii else None

    @property
    def radii(self):
        return self._arrays()[1] if self._radii else None

    @property
    def colours(self):
        return self._arrays()[2] if self._radii else None

    @property
    def material_indecies(self):
        return self._arrays()[3] if self._radii else None

    def _flat(self):
        return dict(type="spheres", centres=self.centres, radii=self.radii, materials=self.material_indecies)

    def get_hits(self, ray_origins, ray_dirs, t_min, t_max):
        """-> (ray_hits, hit_ts, hit_pts, hit_normals, hit_uvs, hit_material_indecies, back_facing).
        Rays without a hit carry zeros, t = t_max + 1 and material 255."""
        ctx = _native.resident_scene("groups", lambda: dict(groups=[self._flat()]))
        r = ctx.intersect(0, ray_origins, ray_dirs, t_min, t_max)
        return r["hit"], r["t"], r["p"], r["n"], r["uv"], r["mat"], r["back"]
