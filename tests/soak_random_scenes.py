"""Randomised parity soak on the GPU box: many generated scenes (the generator of
tests/test_gpu_random_scenes.py with random group counts and sizes), each rendered through the C ABI
with the loops and with every group of 16 or more primitives walked through its hierarchy, every
path compared with the CPU oracle; and the frame's samples rendered in two or three calls (wrt_render,
wrt_render_static), whose accumulation image must equal the oracle's sample sums bit for bit.

    python tests/soak_random_scenes.py [first_seed] [n_scenes] [seconds]

Not collected by pytest (a soak, not a test of fixed length); it lives beside the tests because it uses the
oracle as its checker.

Prints one line per mismatch and a summary; exit status 1 if any path differed.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "raytracing-one-weekend_amd"))
import numpy as np  # noqa: E402

from oracle import numpy_path  # noqa: E402  (the checker; nothing here is a product path)
from tests.test_gpu_random_scenes import Flat, random_scene  # noqa: E402
from weekend_raytracer import main  # noqa: E402


def one_scene(seed):
    rng = np.random.default_rng(seed)
    n_groups = int(rng.integers(1, 14))
    pool = [1, 2, 3, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 20, 30, 31, 33, 40, 64, 65, 100, 150, 257, 400, 700]
    sizes = [int(v) for v in rng.choice(pool, size=int(rng.integers(1, 6)))]
    scene = random_scene(rng, n_groups, sizes)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 40))
    s, b, s_begin = int(rng.integers(1, 5)), int(rng.integers(1, 13)), int(rng.integers(0, 9))
    want = numpy_path.render(scene, w, h, s, b, seed=seed, s_begin=s_begin)
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    bad = 0
    for hierarchy in (0, 16):
        keep = main.HIERARCHY
        main.HIERARCHY = hierarchy
        try:
            got = main.render_frame(objs, w, h, s, b, seed=seed, s_begin=s_begin, want_paths=True)
        finally:
            main.HIERARCHY = keep
        same = np.all(got["path_colours"] == want["path_colours"], axis=1)
        ok = same.all() and np.array_equal(got["segments"], want["segments"]) and \
            np.array_equal(got["counts"], want["counts"])
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} hierarchy {hierarchy}: {w}x{h}x{s} b{b} s_begin {s_begin}, groups {n_groups} "
                  f"sizes {sizes}: {100 * same.mean():.3f} % of paths equal", flush=True)
    # the same samples in two or three calls, through wrt_render and through the entry point that
    # allocates nothing, must leave the same accumulation image as one call (sample by sample addition)
    if s >= 2:
        import torch
        cuts = sorted(set(int(v) for v in rng.integers(1, s, size=2)))
        edges = [0] + cuts + [s]
        frame = main.Frame(objs, hierarchy=0)
        sums = []
        for static in (False, True):
            accum = frame.ctx.zeros((w, h, 3), torch.float32)
            for a, b2 in zip(edges[:-1], edges[1:]):
                (frame.ctx.render_static if static else frame.ctx.render)(w, h, s_begin + a, s_begin + b2, b, seed, accum)
            sums.append(accum.cpu().numpy())
        frame.ctx.close()
        for static, got_sum in zip((False, True), sums):
            if not np.array_equal(got_sum.view(np.uint32), want["sample_sum"].astype(np.single).view(np.uint32)):
                bad += 1
                print(f"MISMATCH seed {seed}: samples in calls {edges}{' (static entry point)' if static else ''} "
                      f"leave another sum than one call", flush=True)
    return bad, w * h * s


def run(first_seed, n_scenes, seconds):
    began = time.time()
    bad = paths = done = 0
    for seed in range(first_seed, first_seed + n_scenes):
        if time.time() - began > seconds:
            break
        b, p = one_scene(seed)
        bad += b
        paths += p
        done += 1
        if done % 20 == 0:
            print(f"{done} scenes, {paths} paths, {bad} mismatches, {time.time() - began:.0f} s", flush=True)
    print(f"soak: {done} scenes (seeds {first_seed}..{first_seed + done - 1}), {paths} paths each way, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    sys.exit(run(args[0] if args else 10000, args[1] if len(args) > 1 else 200, args[2] if len(args) > 2 else 240))
