#!/usr/bin/env python3
"""How the bounce kernel's wavefronts spend their trips in one frame of a workload, from a library built
with -DWRT_TRIP_COUNT=1: trips that shade and with how many lanes, runs of the large
groups' loops (triangles, spheres) and their takers.

    WRT_LIB=raytracing-one-weekend_amd/lib/libwrt_trips.so python tools/trip_counts.py bunny [spp]
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
    out = frame.accumulate(width, height, 0, spp, bounces, 1234)
    torch.cuda.synchronize()
    st = ctx.stats()
    t = st["clock_ticks"].astype(float)
    segments = float(sum(st["rays_per_bounce"][:bounces])) if "rays_per_bounce" in st else float("nan")
    print(f"{name}: {st['wavefronts']} wavefronts, segments {segments:.4g}")
    print(f"  shading trips {t[3]:.4g} with {t[4] / max(1, t[3]):.1f} lanes each ({t[4]:.4g} lane-shadings)")
    print(f"  triangle loop runs {t[5]:.4g} with {t[6] / max(1, t[5]):.1f} takers each, "
          f"sphere loop runs {t[2]:.4g} with {t[7] / max(1, t[2]):.1f} takers each")
    print(f"  shading trips per loop run {t[3] / max(1, t[5] + t[2]):.2f}")
    ctx.close()


if __name__ == "__main__":
    main()
