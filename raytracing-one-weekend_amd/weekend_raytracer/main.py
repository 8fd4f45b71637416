"""
Renderer entry points (reference main.py).  render() keeps the reference's
contract - module-level settings in, {(x, y): rgb} out, y = 0 at the bottom -
and runs the whole bounce loop (reference main.py:93-220) as one persistent
GPU kernel per frame; render_frame() is the array-returning form used by the
benchmark and the tests.
"""
import time

import numpy

from . import _flatten, _native, scenes

IMG_WIDTH = 160
IMG_HEIGHT = 90
ASPECT_RATIO = IMG_WIDTH / IMG_HEIGHT
PIXEL_SAMPLES = 10
MAX_BOUNCES = 10
HORIZON_COLOUR = numpy.array([1.0, 1.0, 1.0], dtype=numpy.single)
SKY_COLOUR = numpy.array([0.5, 0.7, 1.0], dtype=numpy.single)
T_MIN = 0.001
T_MAX = 1000.0
# key of the random streams (the reference uses unseeded generators, main.py:23)
SEED = 0
# Optional, not in the reference: walk large triangle groups through a box hierarchy instead of
# examining every (ray, triangle) pair.  Same image bit for bit, less work.  True: groups of 1024 or
# more triangles; an int: groups of that many or more.  WRT_HIERARCHY in the environment presets it
# (the whole GPU test suite must pass with WRT_HIERARCHY=16 as it does without); None leaves that preset.
HIERARCHY = None

# scene rendered by render(); the reference selects it by editing main.py:79-91
SCENE = scenes.ray_group_triangle_group_bunny_scene


class Frame():
    """A scene resident on one GPU, ready to be rendered repeatedly."""

    def __init__(self, scene, device=None, hierarchy=None):
        camera, groups, material_map = scene
        self.ctx = _native.Context(device)
        hierarchy = HIERARCHY if hierarchy is None else hierarchy
        if hierarchy is not None:
            self.ctx.set_hierarchy(hierarchy)
        self.ctx.upload(_flatten.flatten(camera, groups, material_map, HORIZON_COLOUR, SKY_COLOUR))

    def accumulate(self, width, height, s_begin, s_end, max_bounces, seed, accum=None, paths=None):
        """Trace samples [s_begin, s_end) of every pixel and add their colours
        into accum (float32 CUDA tensor (width, height, 3)); asynchronous."""
        import torch
        if accum is None:
            accum = self.ctx.zeros((width, height, 3), torch.float32)
        self.ctx.render(width, height, s_begin, s_end, max_bounces, seed, accum, T_MIN, T_MAX, paths)
        return accum

    def image(self, accum, spp_total):
        """sqrt(mean over samples): the gamma-2 image of reference main.py:219-220."""
        return self.ctx.finalize(accum, spp_total)


def render_frame(scene, width, height, samples, max_bounces, seed=0, s_begin=0, device=None, want_paths=False):
    """-> dict(image (W,H,3) float32, sample_sum, counts, stats[, path_colours, segments])."""
    import torch
    frame = Frame(scene, device)
    frame.ctx.stats_reset()
    paths = frame.ctx.empty((width * height * samples, 4), torch.float32) if want_paths else None
    accum = frame.accumulate(width, height, s_begin, s_begin + samples, max_bounces, seed, paths=paths)
    image = frame.image(accum, samples)
    stats = frame.ctx.stats()
    out = dict(image=image.cpu().numpy(), sample_sum=accum.cpu().numpy(),
               counts=stats["rays_per_bounce"][:max_bounces + 1], stats=stats)
    if want_paths:
        p = paths.cpu().numpy()
        out["path_colours"] = p[:, :3]
        mark = p[:, 3].astype(numpy.int32)
        out["segments"] = mark % _native.SURVIVOR_MARK
    frame.ctx.close()
    return out


def render():
    """Render SCENE with the module settings; -> {(x, y): float32 rgb}."""
    start = time.perf_counter()
    result = render_frame(SCENE(ASPECT_RATIO), IMG_WIDTH, IMG_HEIGHT, PIXEL_SAMPLES, MAX_BOUNCES, SEED)
    image = result["image"]
    for bounce, rays in enumerate(result["counts"]):
        print(f"Bounce {bounce} of {MAX_BOUNCES}, {rays} rays.")
    pixel_data = {(x, y): image[x, y] for y in range(IMG_HEIGHT) for x in range(IMG_WIDTH)}
    print(f"Total time: {time.perf_counter() - start:0.4f} seconds.")
    return pixel_data


def image_bytes(image):
    """(W,H,3) float image, y up -> (H,W,3) uint8, row 0 at the top, with the
    reference's int(255 * v) quantisation (main.py:57-65): a float32 product,
    truncated, clipped to a byte as PIL does with the fill tuple.  NaN (the
    reference raises on it) becomes 0."""
    flipped = numpy.transpose(numpy.asarray(image, dtype=numpy.single), (1, 0, 2))[::-1]
    scaled = numpy.nan_to_num(numpy.single(255) * flipped, nan=0.0, posinf=255.0, neginf=0.0)
    return numpy.clip(scaled, 0, 255).astype(numpy.int64).astype(numpy.uint8)


def image_bytes_on_device(image, device=None):
    """The same bytes made by the GPU (wrt_image_bytes); image: (W,H,3) numpy array or CUDA tensor."""
    import torch
    ctx = _native.context(device)
    if not isinstance(image, torch.Tensor):
        image = ctx.to_device(image, numpy.single)
    return ctx.image_bytes(image).cpu().numpy()


def generate_image_from_data(img_data, path="tmp_image.png", show=False):
    """Write the pixel dictionary to disk (reference main.py:45-74); pixels absent from it are white."""
    from PIL import Image
    image = numpy.ones((IMG_WIDTH, IMG_HEIGHT, 3), dtype=numpy.single)
    for (x, y), rgb in img_data.items():
        image[x, y] = rgb
    picture = Image.fromarray(image_bytes(image), "RGB")
    if show:
        picture.show()
    picture.save(path)
    return picture


def main():
    print("Start render")
    img_data = render()
    print("End render")
    generate_image_from_data(img_data)


if __name__ == "__main__":
    main()
