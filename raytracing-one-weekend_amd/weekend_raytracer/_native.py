"""
ctypes binding of libwrt.so (C ABI: include/wrt.h) and the device-buffer
plumbing.  PyTorch-ROCm tensors are used only as device memory and streams.

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible, every entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WRT_LIB selects another build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("WRT_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libwrt.so")

MAX_BOUNCES = 62
MAX_MATERIALS = 256
MAX_TEXTURES = 16
SURVIVOR_MARK = 100
STATIC_MAX_PIXELS = 4 << 20

(MAT_NONE, MAT_DIFFUSE, MAT_TEXTURED, MAT_CHECKER, MAT_NORMAL_RGB, MAT_NORMAL_DISCRETE, MAT_METAL,
 MAT_DIELECTRIC) = range(8)


class NativeError(RuntimeError):
    pass


class WrtMaterial(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32), ("texture", ctypes.c_int32),
        ("colour_a", ctypes.c_float * 3), ("colour_b", ctypes.c_float * 3),
        ("scale", ctypes.c_double * 3), ("offset", ctypes.c_double * 3),
        ("fuzziness", ctypes.c_double), ("ior", ctypes.c_double),
    ]


class WrtStats(ctypes.Structure):
    _fields_ = [
        ("rays_per_bounce", ctypes.c_uint64 * (MAX_BOUNCES + 2)),
        ("paths", ctypes.c_uint64), ("sphere_pairs", ctypes.c_uint64), ("triangle_pairs", ctypes.c_uint64),
        ("cull_tests", ctypes.c_uint64), ("disk_pairs", ctypes.c_uint64),
        ("sphere_lane_slots", ctypes.c_uint64), ("triangle_lane_slots", ctypes.c_uint64),
        ("clock_ticks", ctypes.c_uint64 * 8), ("triangle_evaluations", ctypes.c_uint64),
        ("sphere_evaluations", ctypes.c_uint64), ("wavefronts", ctypes.c_uint64),
        ("shelved_paths", ctypes.c_uint64), ("damaged_paths", ctypes.c_uint64),
    ]


_vp, _i, _f, _d, _u64, _i64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_uint64,
                               ctypes.c_int64)
_ip = ctypes.POINTER(ctypes.c_int)

# name -> argument types; every symbol include/wrt.h declares
SIGNATURES = {
    "wrt_create": [_i, ctypes.POINTER(_vp)],
    "wrt_destroy": [_vp],
    "wrt_last_error": [_vp],
    "wrt_device_cus": [_vp],
    "wrt_scene_clear": [_vp],
    "wrt_add_sphere_group": [_vp, _vp, _vp, _vp, _i, _ip],
    "wrt_add_triangle_group": [_vp] + [_vp] * 9 + [_i, _i, _vp, _f, _ip],
    "wrt_add_disk": [_vp, _vp, _vp, _vp, _vp, _d, _i, _ip],
    "wrt_set_materials": [_vp, ctypes.POINTER(WrtMaterial), _i],
    "wrt_set_texture": [_vp, _i, _vp, _i, _i],
    "wrt_set_camera": [_vp] + [_vp] * 6 + [_d],
    "wrt_set_sky": [_vp, _vp, _vp],
    "wrt_scene_commit": [_vp],
    "wrt_render": [_vp, _i, _i, _i, _i, _i, _f, _f, _u64, _vp, _vp, _vp],
    "wrt_render_static": [_vp, _i, _i, _i, _i, _i, _f, _f, _u64, _vp, _vp],
    "wrt_finalize": [_vp, _vp, _i64, _i, _vp, _vp],
    "wrt_image_bytes": [_vp, _vp, _i, _i, _vp, _vp],
    "wrt_profile_enable": [_vp, _i],
    "wrt_profile_read": [_vp, ctypes.POINTER(ctypes.c_double), _ip],
    "wrt_set_triangle_reject": [_vp, _i],
    "wrt_set_path_shelf": [_vp, _i],
    "wrt_set_hierarchy": [_vp, _i],
    "wrt_stats_reset": [_vp, _vp],
    "wrt_stats_read": [_vp, ctypes.POINTER(WrtStats), _vp],
    "wrt_generate_rays": [_vp, _i, _i, _i, _i, _u64, _vp, _vp, _vp],
    "wrt_intersect": [_vp, _i, _vp, _vp, _i64, _f, _f] + [_vp] * 9 + [_vp],
    "wrt_scatter": [_vp, _i] + [_vp] * 8 + [_i, _u64, _i64, _vp, _vp, _vp, _vp],
}

# group size from which the hierarchy walk is faster than the brute-force loops (measured on one MI355X)
HIERARCHY_MIN_TRIANGLES = 1024

_lib = None


def load_library():
    """dlopen libwrt.so and declare every entry point (no GPU needed)."""
    global _lib
    if _lib is None:
        # torch first: its bundled HIP runtime must be the one the process uses
        # (loading the system libamdhip64 before torch leaves torch without devices)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` (hipcc --offload-arch=gfx950)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "wrt_last_error" else ctypes.c_int
        _lib = lib
    return _lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise NativeError("weekend_raytracer needs a ROCm GPU (MI355X); there is no CPU fallback")
    return torch


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.single)


def _hostptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Context:
    """One wrt_ctx on one GPU, plus torch-side helpers."""

    def __init__(self, device=None):
        torch = _torch()
        self.lib = load_library()
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        handle = ctypes.c_void_p()
        rc = self.lib.wrt_create(self.device, ctypes.byref(handle))
        if rc != 0:
            raise NativeError(f"wrt_create: {self.lib.wrt_last_error(None).decode()}")
        self.handle = handle
        preset = int(os.environ.get("WRT_HIERARCHY", "0"))     # see main.HIERARCHY
        if preset:
            self.set_hierarchy(preset)
        # WRT_PATH_SHELF=1 (or 0): see set_path_shelf; the whole GPU test suite must pass with it as it does without
        if "WRT_PATH_SHELF" in os.environ:
            self.set_path_shelf(int(os.environ["WRT_PATH_SHELF"]))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.wrt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self.lib.wrt_last_error(self.handle).decode()}")

    def stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.torch_device).cuda_stream)

    @property
    def compute_units(self):
        return self.lib.wrt_device_cus(self.handle)

    # ---- scene ---------------------------------------------------------
    def upload(self, flat):
        """Upload a flat scene (camera / groups / materials / sky dictionaries;
        any part may be absent)."""
        lib, h = self.lib, self.handle
        self.check(lib.wrt_scene_clear(h), "wrt_scene_clear")
        gid = ctypes.c_int()
        for g in flat.get("groups", []):
            if g["type"] == "spheres":
                c, r = _f32(g["centres"]), _f32(g["radii"])
                m = np.ascontiguousarray(g["materials"], dtype=np.ubyte)
                self.check(lib.wrt_add_sphere_group(h, _hostptr(c), _hostptr(r), _hostptr(m), r.shape[0],
                                                    ctypes.byref(gid)), "wrt_add_sphere_group")
            elif g["type"] == "triangles":
                arrs = [_f32(g[k]) for k in ("pt0", "A", "B", "n0", "n1", "n2", "uv0", "uv1", "uv2")]
                bc = _f32(g["bounds_centre"])
                self.check(lib.wrt_add_triangle_group(h, *[_hostptr(a) for a in arrs], arrs[0].shape[0],
                                                      int(g["material_id"]), _hostptr(bc),
                                                      float(np.single(g["bounds_radius"])), ctypes.byref(gid)),
                           "wrt_add_triangle_group")
            elif g["type"] == "disk":
                vecs = [_f32(g[k]) for k in ("centre", "normal", "U", "V")]
                self.check(lib.wrt_add_disk(h, *[_hostptr(a) for a in vecs], float(g["radius"]),
                                            int(g["material_index"]), ctypes.byref(gid)), "wrt_add_disk")
            else:
                raise ValueError(g["type"])
        mats = flat.get("materials", {})
        if mats:
            count = max(mats) + 1
            if min(mats) < 0 or count > MAX_MATERIALS:
                raise ValueError("material indices must fit the reference's uint8 (0..255)")
            table = (WrtMaterial * count)()
            slot = 0
            for index, m in mats.items():
                e = table[index]
                kind = m["kind"]
                if kind == "diffuse":
                    e.kind = MAT_DIFFUSE
                    e.colour_a[:] = [float(v) for v in _f32(m["colour"])]
                elif kind == "textured":
                    if slot >= MAX_TEXTURES:
                        raise ValueError("too many textures")
                    px = _f32(m["pixels"])
                    self.check(lib.wrt_set_texture(h, slot, _hostptr(px), px.shape[1], px.shape[0]),
                               "wrt_set_texture")
                    e.kind, e.texture = MAT_TEXTURED, slot
                    slot += 1
                elif kind == "checker":
                    e.kind = MAT_CHECKER
                    e.colour_a[:] = [float(v) for v in _f32(m["colour_a"])]
                    e.colour_b[:] = [float(v) for v in _f32(m["colour_b"])]
                    e.scale[:] = [float(v) for v in np.asarray(m["scale"], dtype=np.double)]
                    e.offset[:] = [float(v) for v in np.asarray(m["offset"], dtype=np.double)]
                elif kind == "normal_rgb":
                    e.kind = MAT_NORMAL_RGB
                elif kind == "normal_discrete":
                    e.kind = MAT_NORMAL_DISCRETE
                elif kind == "metal":
                    e.kind = MAT_METAL
                    e.colour_a[:] = [float(v) for v in _f32(m["colour"])]
                    e.fuzziness = float(m["fuzziness"])
                elif kind == "dielectric":
                    e.kind = MAT_DIELECTRIC
                    e.ior = float(m["ior"])
                else:
                    raise ValueError(kind)
            self.check(lib.wrt_set_materials(h, table, count), "wrt_set_materials")
        cam = flat.get("camera")
        if cam is not None:
            v = [np.ascontiguousarray(cam[k], dtype=np.double) for k in ("pos", "U", "V", "bottom_left", "horiz",
                                                                         "vert")]
            self.check(lib.wrt_set_camera(h, *[_hostptr(a) for a in v], float(cam["lens_radius"])),
                       "wrt_set_camera")
        sky = flat.get("sky")
        if sky is not None:
            a, b = _f32(sky["horizon"]), _f32(sky["zenith"])
            self.check(lib.wrt_set_sky(h, _hostptr(a), _hostptr(b)), "wrt_set_sky")
        self.check(lib.wrt_scene_commit(h), "wrt_scene_commit")

    # ---- device helpers ------------------------------------------------
    def to_device(self, a, dtype):
        torch = _torch()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.torch_device)

    def empty(self, shape, dtype):
        torch = _torch()
        return torch.empty(shape, dtype=dtype, device=self.torch_device)

    def zeros(self, shape, dtype):
        torch = _torch()
        return torch.zeros(shape, dtype=dtype, device=self.torch_device)

    # ---- entry points ----------------------------------------------------
    def generate_rays(self, width, height, s_begin, s_end, seed):
        torch = _torch()
        n = width * height * (s_end - s_begin)
        o = self.empty((n, 3), torch.float32)
        d = self.empty((n, 3), torch.float32)
        self.check(self.lib.wrt_generate_rays(self.handle, width, height, s_begin, s_end, seed, _dptr(o), _dptr(d),
                                              self.stream()), "wrt_generate_rays")
        return o, d

    def intersect(self, group, origins, dirs, t_min, t_max):
        """origins, dirs: numpy (n,3).  -> dict of numpy arrays."""
        torch = _torch()
        o = self.to_device(origins, np.single)
        d = self.to_device(dirs, np.single)
        n = o.shape[0]
        out = dict(hit=self.empty((n,), torch.uint8), t=self.empty((n,), torch.float32),
                   p=self.empty((n, 3), torch.float32), n=self.empty((n, 3), torch.float32),
                   uv=self.empty((n, 2), torch.float32), mat=self.empty((n,), torch.uint8),
                   back=self.empty((n,), torch.uint8), group=self.empty((n,), torch.int32),
                   prim=self.empty((n,), torch.int32))
        self.check(self.lib.wrt_intersect(self.handle, group, _dptr(o), _dptr(d), n, t_min, t_max,
                                          *[_dptr(out[k]) for k in ("hit", "t", "p", "n", "uv", "mat", "back",
                                                                    "group", "prim")], self.stream()),
                   "wrt_intersect")
        res = {k: v.cpu().numpy() for k, v in out.items()}
        res["hit"] = res["hit"].astype(bool)
        res["back"] = res["back"].astype(bool)
        return res

    def scatter(self, material_index, dirs, points, normals, uvs, back, px, py, ps, bounce, seed):
        torch = _torch()
        n = np.asarray(dirs).shape[0]
        ins = [self.to_device(a, np.single) for a in (dirs, points, normals, uvs)]
        b = self.to_device(np.asarray(back, dtype=bool).astype(np.ubyte), np.ubyte)
        ids = [self.to_device(np.broadcast_to(np.asarray(a), (n,)), np.int32) for a in (px, py, ps)]
        od, oc = self.empty((n, 3), torch.float32), self.empty((n, 3), torch.float32)
        ab = self.empty((n,), torch.uint8)
        self.check(self.lib.wrt_scatter(self.handle, material_index, *[_dptr(t) for t in ins], _dptr(b),
                                        *[_dptr(t) for t in ids], int(bounce), int(seed), n, _dptr(od), _dptr(oc),
                                        _dptr(ab), self.stream()), "wrt_scatter")
        return od.cpu().numpy(), oc.cpu().numpy(), ab.cpu().numpy().astype(bool)

    def render(self, width, height, s_begin, s_end, max_bounces, seed, accum, t_min=0.001, t_max=1000.0,
               paths=None):
        """Asynchronous on the current torch stream.  accum: float32 CUDA
        tensor (width, height, 3), added to.  paths: optional float32 CUDA
        tensor (width*height*(s_end-s_begin), 4)."""
        self.check(self.lib.wrt_render(self.handle, width, height, s_begin, s_end, max_bounces, t_min, t_max,
                                       int(seed), _dptr(accum), _dptr(paths), self.stream()), "wrt_render")

    def render_static(self, width, height, s_begin, s_end, max_bounces, seed, accum, t_min=0.001, t_max=1000.0):
        """render() through the entry point that allocates nothing: the library's static workspace
        holds the path records (width * height <= STATIC_MAX_PIXELS); same accum bit for bit."""
        self.check(self.lib.wrt_render_static(self.handle, width, height, s_begin, s_end, max_bounces, t_min, t_max,
                                              int(seed), _dptr(accum), self.stream()), "wrt_render_static")

    def finalize(self, accum, spp_total):
        image = self.empty(tuple(accum.shape), accum.dtype)
        self.check(self.lib.wrt_finalize(self.handle, _dptr(accum), accum.numel(), int(spp_total), _dptr(image),
                                         self.stream()), "wrt_finalize")
        return image

    def image_bytes(self, image):
        """(W, H, 3) float32 CUDA tensor, y up -> (H, W, 3) uint8 CUDA tensor, row 0 at the top."""
        torch = _torch()
        width, height = int(image.shape[0]), int(image.shape[1])
        rgb = self.empty((height, width, 3), torch.uint8)
        self.check(self.lib.wrt_image_bytes(self.handle, _dptr(image), width, height, _dptr(rgb), self.stream()),
                   "wrt_image_bytes")
        return rgb

    def profile(self, on):
        self.check(self.lib.wrt_profile_enable(self.handle, 1 if on else 0), "wrt_profile_enable")

    def profile_read(self):
        """-> (summed bounce-kernel milliseconds, launches) since the last read."""
        ms, n = ctypes.c_double(), ctypes.c_int()
        self.check(self.lib.wrt_profile_read(self.handle, ctypes.byref(ms), ctypes.byref(n)), "wrt_profile_read")
        return ms.value, n.value

    def set_triangle_reject(self, on):
        """Bounding-sphere test in front of the triangle evaluation (default on); same results either way."""
        self.check(self.lib.wrt_set_triangle_reject(self.handle, 1 if on else 0), "wrt_set_triangle_reject")

    def set_path_shelf(self, mode):
        """Second path per lane of the bounce kernel, kept in device memory: -1 where it pays (default: scenes
        with two or more large triangle groups), 0 never, 1 in every scene with a triangle group; same
        results in every mode."""
        self.check(self.lib.wrt_set_path_shelf(self.handle, int(mode)), "wrt_set_path_shelf")

    def set_hierarchy(self, on):
        """Walk large triangle groups through a box hierarchy (default off: every pair is examined, as the
        reference does); same closest hits either way, different amount of work.  True: groups of
        HIERARCHY_MIN_TRIANGLES or more; an int: groups of that many triangles or more."""
        n = HIERARCHY_MIN_TRIANGLES if on is True else int(on or 0)
        self.check(self.lib.wrt_set_hierarchy(self.handle, n), "wrt_set_hierarchy")

    def stats_reset(self):
        self.check(self.lib.wrt_stats_reset(self.handle, self.stream()), "wrt_stats_reset")

    def stats(self, strict=True):
        """-> the counters since the last reset.  A render that had to end damaged paths (see wrt_stats.damaged_paths
        in wrt.h; never known to happen) is not to be trusted: NativeError unless strict is False."""
        s = WrtStats()
        self.check(self.lib.wrt_stats_read(self.handle, ctypes.byref(s), self.stream()), "wrt_stats_read")
        if strict and s.damaged_paths:
            raise NativeError(f"the bounce kernel ended {int(s.damaged_paths)} paths whose state named a primitive, "
                              "pixel or sample that does not exist: this render is not to be trusted")
        return dict(damaged_paths=int(s.damaged_paths), rays_per_bounce=np.array(list(s.rays_per_bounce), dtype=np.int64), paths=int(s.paths),
                    sphere_pairs=int(s.sphere_pairs), triangle_pairs=int(s.triangle_pairs),
                    cull_tests=int(s.cull_tests), disk_pairs=int(s.disk_pairs),
                    sphere_lane_slots=int(s.sphere_lane_slots), triangle_lane_slots=int(s.triangle_lane_slots),
                    clock_ticks=np.array(list(s.clock_ticks), dtype=np.uint64).astype(np.int64),
                    triangle_evaluations=int(s.triangle_evaluations), sphere_evaluations=int(s.sphere_evaluations),
                    wavefronts=int(s.wavefronts), shelved_paths=int(s.shelved_paths))


_contexts = {}


def context(device=None):
    """The shared context of a GPU (created on first use)."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def _content_token(value):
    """A digest of everything a flat scene holds (nested dicts / lists of arrays and scalars)."""
    try:
        import xxhash
        h = xxhash.xxh3_128()                      # several GB/s: a texture or a mesh costs a fraction of a millisecond
    except ImportError:
        import hashlib
        h = hashlib.blake2b(digest_size=16)

    def feed(v):
        if isinstance(v, dict):
            for k in sorted(v, key=str):
                h.update(str(k).encode())
                feed(v[k])
        elif isinstance(v, (list, tuple)):
            h.update(b"[%d" % len(v))
            for item in v:
                feed(item)
        elif v is None:
            h.update(b"none")
        else:
            a = np.ascontiguousarray(v)
            h.update(str(a.dtype).encode() + str(a.shape).encode())
            h.update(a.data if a.size else b"")
    feed(value)
    return h.digest()


# The per-object interface (group.get_hits, material.scatter, camera.get_ray_components) keeps what it
# has uploaded resident: a few contexts per GPU and kind of object, least recently used first out, each
# holding one object's flat scene under the digest of its content.  The loop of reference
# main.py:124-200 calls every group and every material once per bounce; with one shared context each of
# those calls would upload its object again (host packing, hipMemcpy, a tree build when the hierarchy
# is on), bounce after bounce.
RESIDENT_CAPACITY = {"groups": 8, "materials": 24, "camera": 2}
_resident = {}
upload_count = 0            # uploads the per-object interface has made in this process


def resident_scene(kind, flat_fn, device=None):
    """-> a context of `device` that holds exactly flat_fn()'s content, uploading it only if no context
    of that kind holds it already.

    The reference's objects are mutable builders (sphere_group.py:25, camera.py:13, any attribute may
    be assigned between calls), so what decides is the content handed to the GPU, not the identity of
    the object: the digest of the flat scene is the key."""
    global upload_count
    from collections import OrderedDict
    torch = _torch()
    device = int(torch.cuda.current_device() if device is None else device)
    held = _resident.setdefault((device, kind), OrderedDict())
    flat = flat_fn()
    token = _content_token(flat)
    ctx = held.get(token)
    if ctx is not None:
        held.move_to_end(token)
        return ctx
    if len(held) >= RESIDENT_CAPACITY[kind]:
        _, ctx = held.popitem(last=False)          # the least recently used context takes the new content
    else:
        ctx = Context(device)
    ctx.upload(flat)
    upload_count += 1
    held[token] = ctx
    return ctx


def release_resident():
    """Close every context the per-object interface holds."""
    for held in _resident.values():
        for ctx in held.values():
            ctx.close()
    _resident.clear()
