"""Randomised soak of the path shelf on the GPU box: generated scenes (the generator of
tests/test_gpu_random_scenes.py: up to 13 groups of up to 700 spheres, triangles or disks, all material
kinds) rendered at frame sizes of several paths per lane of the bounce kernel's resident grid, so that every
wavefront refills and shelves many times, with the shelf forced on (wrt_set_path_shelf 1) and switched off;
every path record, the accumulation image and the pair counts must be equal.  Every so often a scene is
also checked against the CPU oracle at a small size.

    python tests/soak_shelf.py [first_seed] [n_scenes] [seconds]

Not collected by pytest (a soak, not a test of fixed length).  Prints one line per mismatch and a summary;
exit status 1 if anything differed.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "raytracing-one-weekend_amd"))
import numpy as np  # noqa: E402

from tests.test_gpu_random_scenes import Flat, random_scene  # noqa: E402
from weekend_raytracer import main  # noqa: E402


def render(objs, w, h, s, b, seed, s_begin, mode):
    import torch
    frame = main.Frame(objs, hierarchy=0)
    frame.ctx.set_path_shelf(mode)
    frame.ctx.stats_reset()
    paths = torch.full((w * h * s, 4), -7.0, dtype=torch.float32, device="cuda")
    accum = frame.accumulate(w, h, s_begin, s_begin + s, b, seed, paths=paths)
    stats = frame.ctx.stats()
    out = paths.cpu().numpy(), accum.cpu().numpy(), stats
    frame.ctx.close()
    return out


def one_scene(seed):
    rng = np.random.default_rng(seed)
    n_groups = int(rng.integers(2, 14))
    pool = [9, 12, 16, 17, 33, 64, 100, 150, 257, 400, 700]
    sizes = [int(v) for v in rng.choice(pool, size=int(rng.integers(1, 6)))]
    scene = random_scene(rng, n_groups, sizes)
    # 0.4 to 1.6 million paths: one to four paths per lane of the resident grid
    w, h = int(rng.integers(500, 900)), int(rng.integers(300, 500))
    s, b, s_begin = int(rng.integers(1, 5)), int(rng.integers(2, 13)), int(rng.integers(0, 9))
    objs = (Flat(scene["camera"]), [Flat(g) for g in scene["groups"]],
            {i: Flat(m) for i, m in scene["materials"].items()})
    p_on, a_on, st_on = render(objs, w, h, s, b, seed, s_begin, 1)
    p_off, a_off, st_off = render(objs, w, h, s, b, seed, s_begin, 0)
    same = np.all(p_on.view(np.uint32) == p_off.view(np.uint32), axis=1)
    ok = same.all() and np.array_equal(a_on.view(np.uint32), a_off.view(np.uint32)) and \
        np.array_equal(st_on["rays_per_bounce"], st_off["rays_per_bounce"]) and \
        all(st_on[k] == st_off[k] for k in ("triangle_pairs", "sphere_pairs", "cull_tests", "disk_pairs"))
    if not ok:
        print(f"MISMATCH seed {seed}: {w}x{h}x{s} b{b} s_begin {s_begin}, groups {n_groups} sizes {sizes}: "
              f"{100 * same.mean():.4f} % of paths equal", flush=True)
    checked = 0
    if seed % 16 == 0:
        # a corner of the same frame against the oracle (paths do not depend on the frame around them)
        from oracle import numpy_path
        cw, ch = 24, 16
        want = numpy_path.render(scene, cw, ch, s, b, seed=seed, s_begin=s_begin)
        keep = os.environ.get("WRT_PATH_SHELF")
        os.environ["WRT_PATH_SHELF"] = "1"
        try:
            got = main.render_frame(objs, cw, ch, s, b, seed=seed, s_begin=s_begin, want_paths=True)
        finally:
            if keep is None:
                del os.environ["WRT_PATH_SHELF"]
            else:
                os.environ["WRT_PATH_SHELF"] = keep
        if not np.array_equal(got["path_colours"], want["path_colours"]):
            ok = False
            print(f"MISMATCH against the oracle, seed {seed}", flush=True)
        checked = 1
    return ok, int(p_on.shape[0]), int(st_on["shelved_paths"]), checked


def main_():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    began = time.time()
    bad = done = paths = shelved = with_shelf = oracle = 0
    for seed in range(first, first + n):
        if time.time() - began > seconds:
            break
        ok, n_paths, n_shelved, checked = one_scene(seed)
        bad += 0 if ok else 1
        done += 1
        paths += n_paths
        shelved += n_shelved
        with_shelf += 1 if n_shelved else 0
        oracle += checked
        if done % 50 == 0:
            print(f"... {done} scenes, {paths / 1e6:.0f} M paths each way, {shelved / 1e6:.1f} M paths shelved, "
                  f"{bad} mismatches, {time.time() - began:.0f} s", flush=True)
    print(f"soak_shelf: {done} scenes from seed {first}, {paths / 1e6:.0f} M paths each way, shelf used in {with_shelf} "
          f"scenes ({shelved / 1e6:.1f} M paths shelved), {oracle} corners against the oracle, {bad} mismatches, "
          f"{time.time() - began:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main_())
