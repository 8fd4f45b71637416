"""Randomised soak of the optional hierarchy on the GPU box: the shipped meshes (bunny, cow, teapot) and
sphere fields under random similarity transforms (scale 0.01 .. 100, any rotation, offsets up to 2000
units), seen by random cameras (some far away, some inside the meshes' bounds, some skimming past a
vertex, some lying in the plane of a triangle and looking along it), rendered with every group walked and with the uniform loops; every path record
must be equal bit for bit.

    python tests/soak_hierarchy.py [first_seed] [n_scenes] [seconds] [--reject]

--reject: the same configurations through the uniform loops with the conservative per-triangle test
(wrt_set_triangle_reject) switched on and off instead.

Not collected by pytest (a soak, not a test of fixed length).  Exit status 1 if any path differed.
"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "raytracing-one-weekend_amd"))
import numpy as np  # noqa: E402

from oracle import numpy_path  # noqa: E402  (camera basis only)
from tests.test_gpu_reject import Flat  # noqa: E402
from weekend_raytracer import _flatten, main, scenes  # noqa: E402


def base_scenes():
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        for fn in (scenes.ray_group_triangle_group_bunny_scene, scenes.cow_teapot_field_scene, scenes.cloth_bunny_scene):
            out.append(_flatten.flatten(*fn(16 / 9), main.HORIZON_COLOUR, main.SKY_COLOUR))
        out.append(_flatten.flatten(*scenes.cover_scene(16 / 9, scenes.sphere_field_records(480)),
                                    main.HORIZON_COLOUR, main.SKY_COLOUR))
    return out


def rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def transformed(flat, rng):
    scale = float(10.0 ** rng.uniform(-2, 2))
    R = rotation(rng) if rng.random() < 0.8 else np.eye(3)
    offset = rng.uniform(-1, 1, 3) * float(10.0 ** rng.uniform(-1, 3.3))

    def move(p):
        return (np.asarray(p, np.float64) @ R.T) * scale + offset

    groups, centres = [], []
    for g in flat["groups"]:
        g = dict(g)
        if g["type"] == "spheres":
            g["centres"] = move(g["centres"]).astype(np.single)
            g["radii"] = (np.asarray(g["radii"], np.float64) * scale).astype(np.single)
            small = g["radii"] < 50 * scale
            if small.any():
                centres.append(g["centres"][small].astype(np.float64))
        elif g["type"] == "triangles":
            p0 = move(g["pt0"])
            p1 = move(np.asarray(g["pt0"], np.float64) + g["A"])
            p2 = move(np.asarray(g["pt0"], np.float64) + g["B"])
            g["pt0"] = p0.astype(np.single)
            g["A"] = (p1 - p0).astype(np.single)
            g["B"] = (p2 - p0).astype(np.single)
            for k in ("n0", "n1", "n2"):
                g[k] = (np.asarray(g[k], np.float64) @ R.T).astype(np.single)
            pts = np.concatenate([p0, p1, p2])
            bc = pts.mean(axis=0).astype(np.single)
            g["bounds_centre"] = bc
            g["bounds_radius"] = np.single(np.max(np.linalg.norm(pts - bc, axis=1)) * 1.01)
            centres.append(pts)
        else:
            continue                                   # (no disks in these scenes)
        groups.append(g)
    pts = np.concatenate(centres)
    middle, extent = pts.mean(axis=0), float(np.max(np.linalg.norm(pts - pts.mean(axis=0), axis=1)))
    kind = rng.integers(0, 5)
    tri_groups = [g for g in groups if g["type"] == "triangles"]
    if kind == 4 and not tri_groups:
        kind = 3
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    if kind == 0:        # from outside
        pos = middle + direction * extent * float(rng.uniform(1.2, 4.0))
        target = middle + rng.normal(size=3) * 0.2 * extent
    elif kind == 1:      # from far away, narrow view
        pos = middle + direction * extent * float(rng.uniform(8.0, 40.0))
        target = middle
    elif kind == 2:      # from inside the bounds
        pos = middle + rng.normal(size=3) * 0.3 * extent
        target = middle + direction * extent
    elif kind == 4:      # in the plane of one of the triangles (or a hair off it), looking along the plane
        g = tri_groups[int(rng.integers(0, len(tri_groups)))]
        j = int(rng.integers(0, g["pt0"].shape[0]))
        P0, A, B = (np.asarray(g[k][j], np.float64) for k in ("pt0", "A", "B"))
        n = np.cross(A, B)
        n /= max(np.linalg.norm(n), 1e-300)
        a, b2 = rng.uniform(-3, 3, 2)
        lift = float(rng.choice([0.0, 1e-7, 1e-5, 1e-3])) * extent * float(rng.choice([-1.0, 1.0]))
        pos = P0 + a * A + b2 * B + lift * n + (A + B) / max(np.linalg.norm(A + B), 1e-300) * extent * float(rng.uniform(0.0, 1.5))
        target = P0 + float(rng.uniform(0, 1)) * A + float(rng.uniform(0, 1)) * B
        if np.linalg.norm(target - pos) < 1e-6 * extent:
            pos = pos + A * 5.0
    else:                # skimming past a point of the scene
        at = pts[rng.integers(0, len(pts))]
        pos = at + direction * extent * float(rng.uniform(0.5, 2.0))
        side = np.cross(direction, rng.normal(size=3))
        target = at + side / np.linalg.norm(side) * extent * float(rng.uniform(0.0, 0.02))
    fov = float(rng.uniform(3.0, 12.0)) if kind == 1 else float(rng.uniform(20.0, 90.0))
    focus = float(np.linalg.norm(target - pos))
    cam = numpy_path.camera_basis(pos, target, focus, float(rng.choice([0.0, 0.02 * extent])), 16 / 9, fov)
    return dict(flat, camera=cam, groups=groups), scale


def one_scene(bases, seed, what="hierarchy"):
    import torch
    rng = np.random.default_rng(seed)
    scene, scale = transformed(bases[int(rng.integers(0, len(bases)))], rng)
    w, h, s, b = 192, 108, 2, int(rng.integers(2, 9))
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    out, evaluated = [], []          # (evaluated: lane slots of the uniform loops)
    for variant in (True, False):
        # "hierarchy": every group walked against the loops; "reject": the loops with and without the
        # conservative per-triangle test in front of the full evaluation
        frame = main.Frame(objs, hierarchy=16 if (what == "hierarchy" and variant) else 0)
        if what == "reject":
            frame.ctx.set_triangle_reject(variant)
        frame.ctx.stats_reset()
        paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
        frame.accumulate(w, h, 0, s, b, seed, paths=paths)
        st = frame.ctx.stats()
        evaluated.append(int(st["triangle_lane_slots"]) + int(st["sphere_lane_slots"]))
        out.append(paths.cpu().numpy().view(np.uint32))
        if what == "reject":
            evaluated[-1] = -int(st["triangle_evaluations"])      # (rejecting: some pairs reach the full evaluation; off: none is counted)
            frame.ctx.set_triangle_reject(True)
        frame.ctx.close()
    same = np.all(out[0] == out[1], axis=1)
    hits = float(np.mean(out[1][:, 3].view(np.float32) % 100 > 1))
    if not same.all():
        print(f"MISMATCH seed {seed}: scale {scale:.3g}, {100 * same.mean():.4f} % of paths equal", flush=True)
    # walked: fewer lane slots spent in the uniform loops than without the hierarchy
    # (reject: the rejecting loop ran, which is the one that counts the pairs it lets through)
    did_walk = evaluated[0] < evaluated[1]
    return (0 if same.all() else 1), w * h * s, hits, did_walk


def run(first_seed, n_scenes, seconds, what="hierarchy"):
    bases = base_scenes()
    began = time.time()
    bad = paths = done = walked = 0
    bounced = 0.0
    for seed in range(first_seed, first_seed + n_scenes):
        if time.time() - began > seconds:
            break
        b, p, hits, did_walk = one_scene(bases, seed, what)
        bad += b
        paths += p
        bounced += hits
        walked += int(did_walk)
        done += 1
        if done % 20 == 0:
            print(f"{what}: {done} scenes, {paths} paths, {bad} mismatches, {walked} with {'walks' if what == 'hierarchy' else 'rejecting loops'}, "
                  f"{100 * bounced / done:.0f} % of paths hit something, {time.time() - began:.0f} s", flush=True)
    print(f"soak ({what}): {done} scenes (seeds {first_seed}..{first_seed + done - 1}), {paths} paths each way, {walked} scenes "
          f"with {'walks' if what == 'hierarchy' else 'rejecting loops'}, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    what = "reject" if "--reject" in sys.argv else "hierarchy"
    args = [int(v) for v in sys.argv[1:] if not v.startswith("--")]
    sys.exit(run(args[0] if args else 50000, args[1] if len(args) > 1 else 100, args[2] if len(args) > 2 else 240, what))
