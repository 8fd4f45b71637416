"""
Camera: host-side look-at basis (reference camera.py:13-79) and batched
depth-of-field ray generation on the GPU (reference camera.py:108-190).
"""
import math

import numpy

from . import _native

# The reference draws from an unseeded module-level generator (camera.py:5).
# Here every draw is keyed by (SEED, pixel, sample index), so results do not
# depend on call order; S_BEGIN is the global index of sample 0 of the next call.
SEED = 0
S_BEGIN = 0


def set_stream(seed, s_begin=0):
    global SEED, S_BEGIN
    SEED, S_BEGIN = int(seed), int(s_begin)


class Camera():
    def __init__(self, position, lookat, focus_dist, aperture, aspect_ratio, horizontal_fov):
        """position, lookat: numpy 3-vectors; focus_dist: distance to the plane
        in focus; aperture: lens diameter (0 = pinhole); aspect_ratio =
        width / height; horizontal_fov in degrees."""
        width = 2.0 * math.tan(math.radians(horizontal_fov / 2.0))
        height = width / aspect_ratio
        back = position - lookat                       # the camera looks along -W
        self.W = back / numpy.sqrt(back.dot(back))
        right = numpy.cross(numpy.array([0.0, 1.0, 0.0]), self.W)
        self.U = right / numpy.linalg.norm(right)
        self.V = numpy.cross(self.W, self.U)
        self.viewport_horizontal = width * self.U * focus_dist
        self.viewport_vertical = height * self.V * focus_dist
        self.camera_pos = position
        self.bottomleft_focalplane_pos = (self.camera_pos - self.W * focus_dist
                                          - self.viewport_vertical * 0.5 - self.viewport_horizontal * 0.5)
        self.lens_radius = aperture / 2

    def _flat(self):
        return dict(pos=self.camera_pos, U=self.U, V=self.V, W=self.W, horiz=self.viewport_horizontal,
                    vert=self.viewport_vertical, bottom_left=self.bottomleft_focalplane_pos,
                    lens_radius=float(self.lens_radius))

    def get_ray_components(self, width, height, samples):
        """-> (origins, dirs), float32 arrays of shape (width, height, samples, 3)."""
        ctx = _native.resident_scene("camera", lambda: dict(camera=self._flat()))
        o, d = ctx.generate_rays(width, height, S_BEGIN, S_BEGIN + samples, SEED)
        shape = (width, height, samples, 3)
        return o.cpu().numpy().reshape(shape), d.cpu().numpy().reshape(shape)
