"""
Disk primitive: constructor maths of reference disk.py:13-46 on the host,
ray-plane + radius test and planar uvs (reference disk.py:48-94) on the GPU.
"""
import numpy

from . import _native


class Disk():
    def __init__(self, centre, radius, normal, material_index, up=None):
        if up is None:
            self.up = numpy.array([0, 1, 0], dtype=numpy.single)
        else:
            up = numpy.array(up, dtype=numpy.single)
            self.up = (up / numpy.linalg.norm(up)).astype(numpy.single)
        self.centre = centre
        self.radius = radius
        self.radius_squared = radius ** 2
        self.normal = (normal / numpy.linalg.norm(normal)).astype(numpy.single)
        self.material_index = material_index
        # U: to the right, V: to the top, seen from the side the normal points to
        right = numpy.cross(self.up, self.normal)
        self.U = right / numpy.linalg.norm(right)
        self.V = numpy.cross(self.normal, self.U)

    def _flat(self):
        return dict(type="disk", centre=numpy.asarray(self.centre), normal=self.normal, U=self.U, V=self.V,
                    radius=float(self.radius), material_index=int(self.material_index))

    def get_hits(self, ray_origins, ray_dirs, t_min, t_max):
        """-> (hits, ts, hit_pts, hit_normals, hit_uvs, hit_material_indecies, back_facing).
        Entries of rays without a hit are zero (t = t_max + 1); the reference
        leaves unmasked intermediate values there, which main.render() never reads."""
        ctx = _native.resident_scene("groups", lambda: dict(groups=[self._flat()]))
        r = ctx.intersect(0, ray_origins, ray_dirs, t_min, t_max)
        return r["hit"], r["t"], r["p"], r["n"], r["uv"], r["mat"], r["back"]
