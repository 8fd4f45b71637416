"""Pair and segment tallies of one frame of a bench workload (wrt_stats), for cost models:
    python tools/frame_stats.py bunny [cover ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "raytracing-one-weekend_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from weekend_raytracer import main as wr_main  # noqa: E402

for name in sys.argv[1:] or ["bunny"]:
    kind, width, height, spp, bounces, _, _ = bench.WORKLOADS[name]
    spp = int(os.environ.get("WRT_SPP", spp))
    frame = wr_main.Frame(bench.build_scene(kind, width / height), device=0,
                          hierarchy=int(os.environ.get("WRT_HIERARCHY", "0")))
    frame.ctx.stats_reset()
    frame.accumulate(width, height, 0, spp, bounces, 1234)
    torch.cuda.synchronize()
    st = frame.ctx.stats()
    segments = int(st["rays_per_bounce"][:bounces].sum())
    print(f"{name}: paths {st['paths']} segments {segments} wave-segments {segments / 64:.0f}")
    for k in ("sphere_pairs", "triangle_pairs", "cull_tests", "disk_pairs", "sphere_lane_slots", "triangle_lane_slots",
              "triangle_evaluations", "sphere_evaluations"):
        print(f"  {k:22s} {st[k]}")
    frame.ctx.close()
