"""Shared helpers for the tests: fixture loading and comparison metrics."""
import os

import numpy as np

from oracle import scene_io

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ASSETS = os.path.join(REPO, "assets")


def load_texture(name):
    """Decode an asset image the way TexturedDiffuse.__init__ does in the
    reference (materials.py:89-109): /255, drop alpha, flip vertically."""
    from PIL import Image
    img = Image.open(os.path.join(ASSETS, name))
    if img.mode not in ("RGB", "RGBA"):
        raise Exception(f"Unsupported texture image mode: {img.mode}")
    px = np.asarray(img, dtype=np.single).reshape((img.height, img.width, len(img.mode)))
    px = (px / np.single(255.0))[:, :, 0:3]
    return np.ascontiguousarray(np.flipud(px))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_scene(npz):
    return scene_io.unpack_scene(npz, texture_loader=load_texture)


def frame_names():
    return sorted(f[len("frame_"):-4] for f in os.listdir(GOLDEN) if f.startswith("frame_"))


def intersect_names():
    return sorted(f[len("intersect_"):-4] for f in os.listdir(GOLDEN) if f.startswith("intersect_"))


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2))
    return 99.0 if mse == 0.0 else 10.0 * np.log10(1.0 / mse)


def mae(a, b):
    return float(np.mean(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
