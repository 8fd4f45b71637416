"""Where the bounce kernel's time goes: needs a library built with -DWRT_TIMING=1, e.g.

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -DWRT_TIMING=1 \
        raytracing-one-weekend_amd/csrc/wrt.hip -o raytracing-one-weekend_amd/lib/libwrt_timing.so
    WRT_LIB=raytracing-one-weekend_amd/lib/libwrt_timing.so python tools/phase_times.py bunny
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "raytracing-one-weekend_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from weekend_raytracer import main as wr_main  # noqa: E402


def run(name):
    kind, width, height, spp, bounces, _, desc = bench.WORKLOADS[name]
    spp = int(os.environ.get("WRT_SPP", spp))
    # WRT_HIERARCHY=n: groups of n or more triangles are walked through their hierarchies (the walks
    # are tallied with the triangle loops)
    frame = wr_main.Frame(bench.build_scene(kind, width / height), device=0,
                          hierarchy=int(os.environ.get("WRT_HIERARCHY", "0")))
    ctx = frame.ctx
    accum = ctx.zeros((width, height, 3), torch.float32)
    frame.accumulate(width, height, 0, spp, bounces, 1234, accum=accum)
    torch.cuda.synchronize()
    ctx.stats_reset()
    ctx.profile(True)
    ctx.profile_read()
    accum.zero_()
    frame.accumulate(width, height, 0, spp, bounces, 1234, accum=accum)
    torch.cuda.synchronize()
    ms, launches = ctx.profile_read()
    st = ctx.stats()
    t = st["clock_ticks"].astype(float)
    waves = st["wavefronts"] // max(1, launches)
    names = ["sum of wave lifetimes", "longest lifetime", "triangle loops", "sphere loops", "shading", "refill",
             "per-lane walk", "queues"]
    print(f"{name}: kernel {ms:.2f} ms in {launches} launch(es), {waves} waves")
    if t[2:8].sum() == 0:
        print("  no phase tallies: library not built with -DWRT_TIMING=1")
        return
    print(f"  mean lifetime / longest = {t[0] / max(1, st['wavefronts']) / t[1]:.3f}   (ticks per ms = {t[0] / max(1, st['wavefronts']) / (ms / launches):.0f})")
    for i in (2, 3, 4, 5, 6):
        print(f"  {names[i]:22s} {100 * t[i] / t[0]:6.2f} % of wave time")
    print(f"  unaccounted            {100 * (t[0] - t[2:7].sum()) / t[0]:6.2f} %")
    steps = (st["triangle_lane_slots"] / 128, st["sphere_lane_slots"] / 128)
    if steps[0]:
        print(f"  triangle step (2 triangles x 64 lanes): {t[2] / steps[0]:.1f} ticks of one wave")
    if steps[1]:
        print(f"  sphere step (2 spheres x 64 lanes):     {t[3] / steps[1]:.1f} ticks of one wave")
    print(f"  working queues off     {100 * t[7] / t[0]:6.2f} % of wave time (inside the loops)")
    if st["triangle_pairs"]:
        print(f"  triangle pairs fully evaluated: {st['triangle_evaluations'] / st['triangle_pairs']:.4f} of all, "
              f"{st['triangle_evaluations'] / max(1, st['cull_tests']):.2f} per (ray, group)")
    # (walked groups spend no lane slots in loops: with WRT_HIERARCHY set the figures cover the remaining loops' slots only)
    if not int(os.environ.get("WRT_HIERARCHY", "0")):
        tri = f"{st['triangle_pairs'] / st['triangle_lane_slots']:.3f}" if st['triangle_lane_slots'] else "-"
        sph = f"{st['sphere_pairs'] / st['sphere_lane_slots']:.3f}" if st['sphere_lane_slots'] else "-"
        print(f"  lane efficiency of the large groups' loops: triangles {tri}, spheres {sph}")
    ctx.close()


if __name__ == "__main__":
    for n in sys.argv[1:] or ["bunny"]:
        run(n)
