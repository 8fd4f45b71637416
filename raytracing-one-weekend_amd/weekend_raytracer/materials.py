"""
Materials: parameter holders with the reference's constructors
(reference materials.py) whose scatter() runs the GPU scatter kernel.

Random draws are keyed, not sequential: set_stream(seed, px, py, ps, bounce)
names the paths the next scatter() calls belong to (pixel x, y, global sample
index per hit, and the bounce number).  Without it, hit i of call k uses
pixel (i, 0), sample k, bounce 0, so successive calls still differ.

scatter() evaluates what main.render() evaluates: it hands the materials
float64 hit records holding float32 values (main.py:117-120), so reflection,
refraction, Schlick, checker cells and texel indices are computed in float64
and the new direction is rounded to float32 when stored (main.py:198).
"""
import numpy
from PIL import Image

from . import _native

_STREAM = dict(seed=0, px=None, py=None, ps=None, bounce=0, calls=0)


def set_stream(seed, px=None, py=None, ps=None, bounce=0):
    _STREAM.update(seed=int(seed), px=px, py=py, ps=ps, bounce=int(bounce))


def _scatter(material, hit_raydirs, hit_points, hit_normals, hit_uvs, hit_backfaces):
    n = numpy.asarray(hit_raydirs).shape[0]
    st = _STREAM
    px = numpy.arange(n) if st["px"] is None else st["px"]
    py = 0 if st["py"] is None else st["py"]
    ps = st["calls"] if st["ps"] is None else st["ps"]
    st["calls"] += 1
    if n == 0:
        return (hit_points, numpy.zeros((0, 3), dtype=numpy.single), numpy.zeros((0, 3), dtype=numpy.single),
                numpy.zeros((0,), dtype=bool))
    ctx = _native.resident_scene("materials", lambda: dict(materials={0: material._flat()}))
    dirs, cols, absorbed = ctx.scatter(0, hit_raydirs, hit_points, hit_normals, hit_uvs, hit_backfaces,
                                       px, py, ps, st["bounce"], st["seed"])
    return hit_points, dirs, cols, absorbed


class _Material():
    def scatter(self, hit_raydirs, hit_points, hit_normals, hit_uvs, hit_backfaces):
        """-> (new ray origins = hit_points, new directions float32 (n,3),
        colours float32 (n,3), absorbed bool (n,))."""
        return _scatter(self, hit_raydirs, hit_points, hit_normals, hit_uvs, hit_backfaces)


class Diffuse(_Material):
    """Lambertian approximation: bounce towards a random point of the
    hemisphere around the normal (reference materials.py:35-78)."""

    def __init__(self, colour):
        self.colour = colour

    def _flat(self):
        return dict(kind="diffuse", colour=numpy.asarray(self.colour))


class TexturedDiffuse(_Material):
    """Diffuse whose colour is the nearest texel at the hit's uv
    (reference materials.py:81-143).  RGB / RGBA images only."""

    def __init__(self, texture_path):
        texture = Image.open(texture_path)
        channels = {"RGB": 3, "RGBA": 4}
        if texture.mode not in channels:
            raise Exception(f"Unsupported texture image mode: {texture.mode}")
        self.smallest_side = float(min((texture.width, texture.height)))
        pixels = numpy.asarray(texture, dtype=numpy.single).reshape(texture.height, texture.width,
                                                                    channels[texture.mode])
        pixels = pixels / numpy.single(255.0)
        # drop alpha, put row 0 at the bottom
        self.texture_pixels = numpy.flipud(pixels[:, :, 0:3])

    def _flat(self):
        return dict(kind="textured", pixels=numpy.ascontiguousarray(self.texture_pixels),
                    smallest_side=self.smallest_side)


class CheckerboardDiffuse(_Material):
    """Diffuse with a 3-D checker of unit cells in (point * scale + offset)
    space (reference materials.py:146-183)."""

    def __init__(self, scale, offset, colour_a, colour_b):
        self.scale = scale
        self.offset = offset
        self.colour_a = colour_a
        self.colour_b = colour_b

    def _flat(self):
        return dict(kind="checker", scale=numpy.asarray(self.scale), offset=numpy.asarray(self.offset),
                    colour_a=numpy.asarray(self.colour_a), colour_b=numpy.asarray(self.colour_b))


class NormalToRGBDiffuse(_Material):
    """Diffuse coloured (normal + 1) / 2 (reference materials.py:186-207)."""

    def _flat(self):
        return dict(kind="normal_rgb")


class NormalToDiscreteRGBDiffuse(_Material):
    """Diffuse coloured by the axis the normal is within ~37 degrees of, grey
    elsewhere (reference materials.py:14-32, 210-234)."""

    def _flat(self):
        return dict(kind="normal_discrete")


class Metal(_Material):
    """Mirror reflection, optionally perturbed by fuzziness x a random unit
    vector; rays perturbed below the surface are absorbed (reference
    materials.py:237-295).  The reference's fuzzy branch raises NameError as
    shipped (materials.py:274-275); this implements what those lines intend."""

    def __init__(self, colour, fuzziness):
        self.colour = colour
        self.fuzziness = fuzziness

    def _flat(self):
        return dict(kind="metal", colour=numpy.asarray(self.colour), fuzziness=float(self.fuzziness))


class Dielectric(_Material):
    """Glass: Snell refraction, total internal reflection and Schlick's
    reflectance (reference materials.py:298-398)."""

    def __init__(self, index_of_refraction):
        self.ior = index_of_refraction

    def _flat(self):
        return dict(kind="dielectric", ior=float(self.ior))
