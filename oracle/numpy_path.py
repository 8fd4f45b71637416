"""
TEST INFRASTRUCTURE - CPU restatement (numpy) of the reference's vectorised
render path, working on flat scenes (oracle/scene_io.py) and keyed random
streams (oracle/keyed_rng.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the shipped renderer never does.  Pinned against the
fixtures that oracle/gen_golden.py produced by running the reference itself
(tests/test_oracle_golden.py): every function below must reproduce them.

Like the reference it evaluates dense (rays x primitives) grids, so its
speed is representative of the reference's numpy path; rays are processed in
fixed-size chunks (the reference sizes its chunks from free host memory,
sphere_group.py:47-70, triangle_group.py:121-147; chunking is result-neutral).
"""
import numpy as np

from . import keyed_rng

T_MIN = 0.001
T_MAX = 1000.0
PAIR_BUDGET = 3_000_000      # (rays x primitives) pairs evaluated per chunk


def path_identity(width, height, samples, s_begin=0):
    """Ray id r = (x*H + y)*S + s (camera.py:121-128, main.py:102-104)."""
    ids = np.arange(width * height * samples)
    return ids // (height * samples), (ids // samples) % height, ids % samples + s_begin


# --------------------------------------------------------------------------
# camera (camera.py:13-79, 108-190)
# --------------------------------------------------------------------------
def camera_basis(position, lookat, focus_dist, aperture, aspect_ratio, horizontal_fov):
    import math
    viewport_width = math.tan(math.radians(horizontal_fov / 2.0)) * 2.0
    viewport_height = viewport_width / aspect_ratio
    w_dir = position - lookat
    W = w_dir / np.sqrt(w_dir.dot(w_dir))
    U = np.cross(np.array([0.0, 1.0, 0.0]), W)
    U = U / np.linalg.norm(U)
    V = np.cross(W, U)
    horiz = viewport_width * U * focus_dist
    vert = viewport_height * V * focus_dist
    bottom_left = position - W * focus_dist - vert * 0.5 - horiz * 0.5
    return dict(pos=position, U=U, V=V, W=W, horiz=horiz, vert=vert, bottom_left=bottom_left,
                lens_radius=aperture / 2)


def generate_rays(cam, width, height, samples, seed, s_begin=0):
    """-> origins, dirs: (W*H*S, 3) float32 in ray-id order."""
    x, y, s = path_identity(width, height, samples, s_begin)
    return generate_rays_for(cam, width, height, x, y, s, seed)


def generate_rays_for(cam, width, height, x, y, s, seed):
    """Camera rays of the paths named by pixel (x, y) and global sample index s."""
    n = x.shape[0]
    jitter = keyed_rng.jitter(seed, x, y, s)
    # float32 pixel index + float64 offset, stored float32; then a float32 divide
    vp_u = (x.astype(np.single).astype(np.double) + jitter[:, 0]).astype(np.single) / np.single(width)
    vp_v = (y.astype(np.single).astype(np.double) + jitter[:, 1]).astype(np.single) / np.single(height)

    disk = np.zeros((n, 3))
    pending = np.arange(n)
    attempt = 0
    while pending.size:
        cand = np.zeros((pending.size, 3))
        cand[:, 0:2] = keyed_rng.lens_candidate(seed, x[pending], y[pending], s[pending], attempt)
        disk[pending] = cand
        pending = pending[np.einsum("ij,ij->i", cand, cand) > 1.0]
        attempt += 1

    offset = cam["lens_radius"] * disk
    oriented = offset[:, 0, np.newaxis] * cam["U"] + offset[:, 1, np.newaxis] * cam["V"]
    on_plane = (cam["bottom_left"] + cam["horiz"] * vp_u[:, np.newaxis]
                + cam["vert"] * vp_v[:, np.newaxis])
    origins = oriented + cam["pos"]
    dirs = on_plane - origins
    dirs /= np.sqrt(np.einsum("ij,ij->i", dirs, dirs))[:, np.newaxis]
    return origins.astype(np.single), dirs.astype(np.single)


# --------------------------------------------------------------------------
# intersection
# --------------------------------------------------------------------------
def _chunks(n, prims):
    step = max(1, PAIR_BUDGET // max(1, prims))
    for lo in range(0, n, step):
        yield slice(lo, min(n, lo + step))


def _empty_record(n, t_fill, material):
    return dict(hit=np.zeros(n, dtype=bool), t=np.full(n, t_fill),
                p=np.zeros((n, 3), dtype=np.single), n=np.zeros((n, 3), dtype=np.single),
                uv=np.zeros((n, 2), dtype=np.single), mat=np.full(n, material, dtype=np.ubyte),
                back=np.zeros(n, dtype=bool), prim=np.full(n, -1, dtype=np.int32))


def _spheres_chunk(g, o, d, t_min, t_max):
    """sphere_group.py:124-261."""
    centres, radii = g["centres"], g["radii"]
    n = o.shape[0]
    sentinel = np.single(t_max + 1.0)
    co = o[:, np.newaxis] - centres
    h = np.einsum("...i,...ki", d, co)
    c = np.einsum("...ij,...ij->...i", co, co) - radii ** 2
    del co
    disc = np.square(h) - c
    ok = disc > 0.00001
    root = np.sqrt(disc[ok])
    near = np.full_like(disc, sentinel)
    near[ok] = -h[ok] - root
    far = np.full_like(disc, sentinel)
    far[ok] = -h[ok] + root
    near[near < t_min] = sentinel
    far[far < t_min] = sentinel
    best = np.minimum(near, far)
    prim = np.argmin(best, axis=1)
    t = best[np.arange(n), prim]
    hit = t < t_max
    p = np.zeros((n, 3), dtype=np.single)
    p[hit] = o[hit] + d[hit] * t[hit][..., np.newaxis]
    which = np.where(hit, prim, -1)
    nrm = np.zeros((n, 3), dtype=np.single)
    nrm[hit] = (p[hit] - centres[which[hit]]) / radii[which[hit]][..., np.newaxis]
    back = np.einsum("ij,ij->i", nrm, d) > 0.0
    nrm[back] *= -1.0
    mat = np.where(hit, g["materials"][which], 255).astype(np.ubyte)
    return dict(hit=hit, t=t, p=p, n=nrm, uv=np.zeros((n, 2), dtype=np.single), mat=mat, back=back,
                prim=which.astype(np.int32))


def hit_spheres(g, o, d, t_min=T_MIN, t_max=T_MAX):
    parts = [_spheres_chunk(g, o[c], d[c], t_min, t_max) for c in _chunks(o.shape[0], g["radii"].shape[0])]
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]} if parts else \
        _empty_record(0, t_max + 1.0, 255)


def _triangles_chunk(g, o, d, t_min, t_max):
    """triangle_group.py:267-452 (Moller-Trumbore, every pair fully evaluated)."""
    A, B, P0 = g["A"], g["B"], g["pt0"]
    n = o.shape[0]
    rows = np.arange(n)
    pvec = np.cross(d[:, np.newaxis], B[np.newaxis, :])
    det = np.einsum("ij,...ij->...i", A, pvec)
    parallel = np.absolute(det) < 0.00001
    inv = det.copy()
    inv[parallel] = 1.0
    inv = 1.0 / inv
    tvec = o[:, np.newaxis] - P0
    u = np.einsum("...ij,...ij->...i", tvec, pvec) * inv
    del pvec
    qvec = np.cross(tvec, A[np.newaxis, :])
    del tvec
    v = np.einsum("...j,...ij->...i", d, qvec) * inv
    t = np.einsum("ij,...ij->...i", B, qvec) * inv
    del qvec
    miss = parallel | (u > 1.0) | (u < 0.0) | (v < 0.0) | ((u + v) > 1.0) | (t < t_min) | (t > t_max)
    t[miss] = t_max + 1
    prim = np.argmin(t, axis=1)
    best_t = t[rows, prim]
    hit = best_t < t_max
    p = np.zeros((n, 3), dtype=np.single)
    p[hit] = o[hit] + d[hit] * best_t[hit][..., np.newaxis]
    uh = u[hit, prim[hit], np.newaxis]
    vh = v[hit, prim[hit], np.newaxis]
    nrm = np.zeros((n, 3), dtype=np.single)
    nrm[hit] = g["n0"][prim[hit]] * (1.0 - uh - vh) + g["n1"][prim[hit]] * uh + g["n2"][prim[hit]] * vh
    nrm[hit] /= np.linalg.norm(nrm[hit], axis=1)[:, np.newaxis]
    uv = np.zeros((n, 2), dtype=np.single)
    uv[hit] = g["uv0"][prim[hit]] * (1.0 - uh - vh) + g["uv1"][prim[hit]] * uh + g["uv2"][prim[hit]] * vh
    back = det[rows, prim] < 0.0
    nrm[back] *= -1.0
    return dict(hit=hit, t=best_t, p=p, n=nrm, uv=uv, mat=np.full(n, g["material_id"], dtype=np.ubyte),
                back=back, prim=np.where(hit, prim, -1).astype(np.int32))


def cull_mask(g, o, d):
    """Bounding-sphere pre-cull, triangle_group.py:158-166 (no t test)."""
    co = o - g["bounds_centre"]
    h = np.einsum("ij,ij->i", d, co)
    c = np.einsum("ij,ij->i", co, co) - g["bounds_radius"] ** 2
    return (h ** 2 - c) > 0.0001


def hit_triangles(g, o, d, t_min=T_MIN, t_max=T_MAX):
    """triangle_group.py:150-264: cull, evaluate survivors, scatter back."""
    n = o.shape[0]
    out = _empty_record(n, t_max + 1, g["material_id"])
    keep = cull_mask(g, o, d)
    if not np.any(keep):
        return out
    oo, dd = o[keep], d[keep]
    parts = [_triangles_chunk(g, oo[c], dd[c], t_min, t_max) for c in _chunks(oo.shape[0], g["A"].shape[0])]
    for k in out:
        out[k][keep] = np.concatenate([p[k] for p in parts])
    return out


def hit_disk(g, o, d, t_min=T_MIN, t_max=T_MAX):
    """disk.py:48-94."""
    n = o.shape[0]
    normal, centre = g["normal"], g["centre"]
    d_dot_n = np.einsum("ij,j->i", d, normal)
    c_minus_o = centre - o
    num = np.einsum("ij,j->i", c_minus_o, normal)
    hit = np.abs(d_dot_n) > 0.0001
    t = np.full(n, t_max + 1.0, dtype=np.single)
    t[hit] = num[hit] / d_dot_n[hit]
    hit = hit & (t > t_min) & (t < t_max)
    p = o + d * t[:, np.newaxis]
    rel = p - centre
    hit = hit & (np.einsum("ij,ij->i", rel, rel) < g["radius"] ** 2)
    nrm = np.full((n, 3), normal, dtype=np.single)
    rel = rel.astype(np.single)
    u = (np.einsum("ij,j->i", rel, g["U"] / g["radius"]) + 1.0) / 2.0
    v = (np.einsum("ij,j->i", rel, g["V"] / g["radius"]) + 1.0) / 2.0
    back = d_dot_n > 0.0
    nrm[back] *= -1.0
    return dict(hit=hit, t=t, p=p, n=nrm, uv=np.column_stack((u, v)),
                mat=np.full(n, g["material_index"], dtype=np.ubyte), back=back,
                prim=np.where(hit, 0, -1).astype(np.int32))


def hit_group(g, o, d, t_min=T_MIN, t_max=T_MAX):
    return {"spheres": hit_spheres, "triangles": hit_triangles, "disk": hit_disk}[g["type"]](g, o, d, t_min, t_max)


def closest_hit(groups, o, d, t_min=T_MIN, t_max=T_MAX):
    """Merge over groups, main.py:116-173: a later group replaces the record
    only when strictly closer.  Records are float64 arrays of float32 values."""
    n = o.shape[0]
    rec = dict(hit=np.zeros(n, dtype=bool), t=np.full(n, 5001.0), p=np.zeros((n, 3)), n=np.zeros((n, 3)),
               uv=np.zeros((n, 2)), mat=np.full(n, 255, dtype=np.ubyte), back=np.zeros(n, dtype=bool),
               prim=np.full(n, -1, dtype=np.int32), group=np.full(n, -1, dtype=np.int32))
    for gi, g in enumerate(groups):
        cand = hit_group(g, o, d, t_min, t_max)
        closer = cand["hit"] & (cand["t"] < rec["t"])
        rec["hit"] |= cand["hit"]
        for k in ("t", "mat", "back", "prim"):
            rec[k] = np.where(closer, cand[k], rec[k])
        for k in ("p", "n", "uv"):
            rec[k] = np.where(closer[:, np.newaxis], cand[k], rec[k])
        rec["group"] = np.where(closer, gi, rec["group"])
    return rec


# --------------------------------------------------------------------------
# materials (materials.py)
# --------------------------------------------------------------------------
AXIS_COLOURS = [  # materials.py:14-32, in evaluation order
    ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0)), ((0.0, 1.0, 0.0), (0.0, 1.0, 0.0)), ((0.0, 0.0, 1.0), (0.0, 0.0, 1.0)),
    ((-1.0, 0.0, 0.0), (1.0, 0.0, 1.0)), ((0.0, -1.0, 0.0), (1.0, 1.0, 0.0)), ((0.0, 0.0, -1.0), (0.0, 1.0, 1.0)),
]


def _reflect(dirs, normals):
    return dirs - (normals * 2.0 * np.einsum("ij,ij->i", dirs, normals)[..., np.newaxis])


def _hemisphere(seed, ident, bounce, normals):
    vec = keyed_rng.keyed_unit_vectors(seed, ident[0], ident[1], ident[2], bounce)
    wrong = np.einsum("ij,ij->i", vec, normals) < 0.0
    vec[wrong] *= -1.0
    return vec


def scatter(mat, dirs, points, normals, uvs, back, seed, ident, bounce):
    """-> (new directions [float64 or float32 as the reference returns],
           colours, absorbed).  The new origin is the hit point.
    ident = (x, y, sample) arrays naming the path of every hit."""
    n = dirs.shape[0]
    kind = mat["kind"]
    absorbed = np.zeros(n, dtype=bool)
    if kind in ("diffuse", "textured", "checker", "normal_rgb", "normal_discrete"):
        out = _hemisphere(seed, ident, bounce, normals)
        if kind == "diffuse":
            col = np.full((n, 3), mat["colour"], dtype=np.single)
        elif kind == "textured":
            idx = (np.clip(uvs, 0.0, 1.0) * (mat["smallest_side"] - 1.0)).astype(np.intc)
            col = mat["pixels"][idx[:, 1], idx[:, 0]]
        elif kind == "checker":
            cell = [np.remainder(np.fabs(np.floor(points[:, a] * mat["scale"][a] + mat["offset"][a])), 2)
                    for a in range(3)]
            pick = np.logical_xor(cell[0], np.logical_xor(cell[1], cell[2]))
            col = np.where(pick[:, np.newaxis], mat["colour_a"], mat["colour_b"])
        elif kind == "normal_rgb":
            col = (normals + 1.0) * 0.5
        else:
            col = np.full((n, 3), [0.4, 0.4, 0.4], dtype=np.single)
            for axis, colour in AXIS_COLOURS:
                col[np.einsum("j,ij->i", np.array(axis), normals) > 0.8] = colour
        return out, col, absorbed
    if kind == "metal":
        out = _reflect(dirs, normals)
        col = np.full((n, 3), mat["colour"], dtype=np.single)
        if mat["fuzziness"] > 0.0001:
            # intended semantics of materials.py:273-292 (the shipped lines 274-275 raise NameError)
            out = out + keyed_rng.keyed_unit_vectors(seed, ident[0], ident[1], ident[2], bounce) * mat["fuzziness"]
            out /= np.sqrt(np.einsum("ij,ij->i", out, out))[..., np.newaxis]
            inside = np.einsum("ij,ij->i", out, normals) < 0.00001
            col[inside] = 0.0
            absorbed[inside] = True
        return out, col, absorbed
    if kind == "dielectric":
        ratio = np.full(n, mat["ior"], dtype=np.single)
        ratio = np.where(np.logical_not(back), 1.0 / ratio, ratio)
        cos_t = np.minimum(np.einsum("ij,ij->i", (-1.0 * dirs), normals), 1.0)
        sin_t = np.sqrt(1.0 - cos_t ** 2)
        cannot = (ratio * sin_t) > 1.0
        r0 = ((1.0 - ratio) / (1.0 + ratio)) ** 2
        reflectance = r0 + ((1.0 - r0) * ((1.0 - cos_t) ** 5))
        coin = keyed_rng.keyed_schlick(seed, ident[0], ident[1], ident[2], bounce)
        reflect = np.logical_or(cannot, reflectance > coin)
        refract = np.logical_not(reflect)
        out = np.zeros((n, 3), dtype=np.single)
        out[reflect] = _reflect(dirs[reflect], normals[reflect])
        perp = ratio[refract][..., np.newaxis] * (dirs[refract] + cos_t[refract][..., np.newaxis] * normals[refract])
        par = (-1.0 * np.sqrt(np.abs(1.0 - np.einsum("ij,ij->i", perp, perp))))[..., np.newaxis] * normals[refract]
        out[refract] = perp + par
        return out, np.ones((n, 3), dtype=np.single), absorbed
    raise ValueError(kind)


# --------------------------------------------------------------------------
# whole frame (main.py:93-220)
# --------------------------------------------------------------------------
def render(scene, width, height, samples, max_bounces, seed, s_begin=0, t_min=T_MIN, t_max=T_MAX,
           trace=None):
    """-> dict(image (W,H,3) sqrt-gamma, sample_sum (W,H,3), path_colours (N,3),
               counts (B+1,) active rays entering each bounce, segments (N,))."""
    x, y, s = path_identity(width, height, samples, s_begin)
    r = trace_paths(scene, width, height, x, y, s, max_bounces, seed, t_min, t_max, trace)
    stacked = r["path_colours"].reshape(width, height, samples, 3)
    r.update(image=np.sqrt(np.mean(stacked, axis=2)), sample_sum=np.sum(stacked, axis=2, dtype=np.single))
    return r


def trace_paths(scene, width, height, x, y, s, max_bounces, seed, t_min=T_MIN, t_max=T_MAX, trace=None):
    """The bounce loop for an arbitrary set of paths (pixel x, y and global
    sample index s per path) of a width x height image.
    -> dict(path_colours (n,3), counts (B+1,), segments (n,))."""
    o, d = generate_rays_for(scene["camera"], width, height, x, y, s, seed)
    n = o.shape[0]
    colours = np.ones((n, max_bounces + 1, 3), dtype=np.single)
    horizon, zenith = scene["sky"]["horizon"], scene["sky"]["zenith"]
    active = np.arange(n)
    counts = []
    segments = np.zeros(n, dtype=np.int32)
    for bounce in range(max_bounces + 1):
        counts.append(active.shape[0])
        if bounce == max_bounces:
            colours[active, bounce] = 0.0
            break
        segments[active] += 1
        in_o, in_d = o[active], d[active]
        rec = closest_hit(scene["groups"], in_o, in_d, t_min, t_max)
        absorbed = np.zeros(active.shape[0], dtype=bool)
        for index, mat in scene["materials"].items():
            m = (rec["mat"] == index) & rec["hit"]
            sel = active[m]
            new_d, col, ab = scatter(mat, d[sel], rec["p"][m], rec["n"][m], rec["uv"][m], rec["back"][m],
                                     seed, (x[sel], y[sel], s[sel]), bounce)
            o[sel] = rec["p"][m]
            d[sel] = new_d
            colours[sel, bounce] = col
            absorbed[m] = ab
        miss = active[~rec["hit"]]
        ts = (d[miss, 1] + 1.0) * 0.5
        colours[miss, bounce] = (1.0 - ts)[..., np.newaxis] * horizon + ts[..., np.newaxis] * zenith
        if trace is not None:
            trace.append(dict(bounce=bounce, active=active.copy(), in_o=in_o, in_d=in_d, rec=rec,
                              out_o=o[active].copy(), out_d=d[active].copy(),
                              colour=colours[active, bounce].copy(), absorbed=absorbed.copy()))
        active = active[rec["hit"] & ~absorbed]
    return dict(path_colours=np.prod(colours, axis=1), counts=np.array(counts, dtype=np.int64), segments=segments)
