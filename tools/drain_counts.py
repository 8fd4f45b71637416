#!/usr/bin/env python3
"""The pooled triangle drain's work in one frame of a workload, from a library built with
-DWRT_DRAIN_COUNT=1 (WRT_LIB=.../libwrt_cnt.so): drains, expansion trips, items, evaluation rounds.

    WRT_LIB=raytracing-one-weekend_amd/lib/libwrt_cnt.so python tools/drain_counts.py bunny [spp]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "raytracing-one-weekend_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402


def main():
    import torch
    from weekend_raytracer import main as wr_main
    name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    kind, width, height, spp, bounces, _, _ = bench.WORKLOADS[name]
    if len(sys.argv) > 2:
        spp = int(sys.argv[2])
    frame = wr_main.Frame(bench.build_scene(kind, width / height))
    ctx = frame.ctx
    ctx.stats_reset()
    frame.accumulate(width, height, 0, spp, bounces, 1234)
    torch.cuda.synchronize()
    st = ctx.stats()
    t = st["clock_ticks"]
    loops = st["triangle_lane_slots"] / 64.0
    groups = {}
    print(f"{name}: {st['wavefronts']} wavefronts, triangle pairs {st['triangle_pairs']:.4g}, evaluated {st['triangle_evaluations']:.4g}")
    print(f"  drains (all kinds) {t[2]}, expansion trips {t[3]}, items {t[4]}, rounds {t[5]}, sum of longest lists {t[6]}")
    if t[2]:
        print(f"  per pooled drain: trips {t[3] / t[2]:.1f}, items {t[4] / t[2]:.1f}, rounds {t[5] / t[2]:.2f}, longest list {t[6] / t[2]:.1f}; "
              f"items per round {t[4] / max(1, t[5]):.1f}")
    ctx.close()


if __name__ == "__main__":
    main()
