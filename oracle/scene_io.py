"""
TEST INFRASTRUCTURE - flat scene records and their .npz (de)serialisation.

A "flat scene" is the neutral description every checker in this repository
exchanges: plain numpy arrays, no classes.

    scene = {
      "camera":   {"pos","U","V","W","horiz","vert","bottom_left": (3,) float64,
                   "lens_radius": float},
      "groups":   [ {"type": "spheres",   "centres" (K,3) f32, "radii" (K,) f32,
                     "materials" (K,) u8},
                    {"type": "triangles", "pt0","A","B","n0","n1","n2" (F,3) f32,
                     "uv0","uv1","uv2" (F,2) f32, "material_id" int,
                     "bounds_centre" (3,) f32, "bounds_radius" f32},
                    {"type": "disk", "centre","normal","U","V" (3,) f32,
                     "radius" float, "material_index" int} ... ],   # scene order
      "materials": {index: {"kind": ..., parameters...}},            # map order
      "sky":      {"horizon" (3,) f32, "zenith" (3,) f32},
    }

Material kinds and parameters:
    diffuse(colour) textured(pixels (h,w,3) f32 already flipped, smallest_side)
    checker(scale, offset, colour_a, colour_b) normal_rgb() normal_discrete()
    metal(colour, fuzziness) dielectric(ior)
"""
import hashlib
import json

import numpy as np

GROUP_FIELDS = {
    "spheres": ["centres", "radii", "materials"],
    "triangles": ["pt0", "A", "B", "n0", "n1", "n2", "uv0", "uv1", "uv2",
                  "material_id", "bounds_centre", "bounds_radius"],
    "disk": ["centre", "normal", "U", "V", "radius", "material_index"],
}
MATERIAL_FIELDS = {
    "diffuse": ["colour"],
    "textured": ["pixels", "smallest_side"],
    "checker": ["scale", "offset", "colour_a", "colour_b"],
    "normal_rgb": [],
    "normal_discrete": [],
    "metal": ["colour", "fuzziness"],
    "dielectric": ["ior"],
}
CAMERA_FIELDS = ["pos", "U", "V", "W", "horiz", "vert", "bottom_left", "lens_radius"]


def texture_digest(pixels):
    return hashlib.sha256(np.ascontiguousarray(pixels, dtype=np.single).tobytes()).hexdigest()


def pack_scene(scene, prefix="scene_", texture_files=None):
    """Flat scene -> {key: array} suitable for numpy.savez.  If a textured
    material's index is in texture_files ({index: asset file name}) only the
    file name and a digest of the decoded pixels are stored, not the pixels."""
    texture_files = texture_files or {}
    out = {}
    meta = {"groups": [], "materials": []}
    for f in CAMERA_FIELDS:
        out[f"{prefix}cam_{f}"] = np.asarray(scene["camera"][f], dtype=np.float64)
    for gi, grp in enumerate(scene["groups"]):
        meta["groups"].append(grp["type"])
        for f in GROUP_FIELDS[grp["type"]]:
            out[f"{prefix}g{gi}_{f}"] = np.asarray(grp[f])
    for index, mat in scene["materials"].items():
        entry = {"index": int(index), "kind": mat["kind"]}
        for f in MATERIAL_FIELDS[mat["kind"]]:
            if f == "pixels" and index in texture_files:
                entry["texture_file"] = texture_files[index]
                entry["texture_sha256"] = texture_digest(mat["pixels"])
                entry["texture_shape"] = list(mat["pixels"].shape)
                continue
            out[f"{prefix}m{int(index)}_{f}"] = np.asarray(mat[f])
        meta["materials"].append(entry)
    out[f"{prefix}sky_horizon"] = np.asarray(scene["sky"]["horizon"], dtype=np.single)
    out[f"{prefix}sky_zenith"] = np.asarray(scene["sky"]["zenith"], dtype=np.single)
    out[f"{prefix}meta"] = np.asarray(json.dumps(meta))
    return out


def unpack_scene(npz, prefix="scene_", texture_loader=None):
    """Inverse of pack_scene.  texture_loader(file name) -> (h,w,3) f32 flipped
    pixels is needed when the fixture stores textures by file name; the
    digest recorded in the fixture is checked."""
    meta = json.loads(str(npz[f"{prefix}meta"]))
    cam = {}
    for f in CAMERA_FIELDS:
        v = npz[f"{prefix}cam_{f}"]
        cam[f] = float(v) if v.ndim == 0 else v
    groups = []
    for gi, gtype in enumerate(meta["groups"]):
        grp = {"type": gtype}
        for f in GROUP_FIELDS[gtype]:
            v = npz[f"{prefix}g{gi}_{f}"]
            if f in ("material_id", "material_index"):
                v = int(v)
            elif f == "radius":
                v = float(v)
            grp[f] = v
        groups.append(grp)
    materials = {}
    for entry in meta["materials"]:
        index = entry["index"]
        mat = {"kind": entry["kind"]}
        for f in MATERIAL_FIELDS[entry["kind"]]:
            if f == "pixels" and "texture_file" in entry:
                if texture_loader is None:
                    raise ValueError("fixture stores a texture by name; pass texture_loader")
                pixels = texture_loader(entry["texture_file"])
                if texture_digest(pixels) != entry["texture_sha256"]:
                    raise ValueError(f"decoded texture {entry['texture_file']} differs from the fixture's")
                mat["pixels"] = pixels
                continue
            v = npz[f"{prefix}m{index}_{f}"]
            if f in ("fuzziness", "ior", "smallest_side"):
                v = float(v)
            mat[f] = v
        materials[index] = mat
    sky = {"horizon": npz[f"{prefix}sky_horizon"], "zenith": npz[f"{prefix}sky_zenith"]}
    return {"camera": cam, "groups": groups, "materials": materials, "sky": sky}
