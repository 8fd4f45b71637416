#!/usr/bin/env python3
"""
Benchmark of the render hot path: Mrays/s counting every bounce segment
(sum over bounces b < max_bounces of the rays entering bounce b, the counts the
reference prints at main.py:111) divided by wall time per frame.

    python bench.py --gpus N --steps K --warmup W [--workload cover|bunny|cloth|demo|cow_teapot]

One "step" is one full frame of the workload: the persistent bounce kernel
over all pixels x samples, the per-pixel sample reduction, (N > 1) the RCCL
all-reduce of the float32 accumulation images, and the sqrt/mean finalise.
The scene is resident in HBM before timing starts.  With N > 1 every rank
renders its own samples of every pixel (disjoint global sample indices) and the
images are summed over xGMI: `--scaling weak` (default) gives every rank the
workload's spp, `--scaling strong` splits the workload's spp over the ranks.

Started without torch.distributed.run (`python bench.py --gpus N`, N > 1), the
process starts N child ranks of itself before it touches the GPU, relays rank
0's JSON line and exits with the children's status.

Rank 0 prints ONE JSON line.  `roofline` prices the bounce kernel against the
FP32 vector peak with the algorithmic flop count of the reference's dense
evaluation (45 per ray-triangle pair after the cull, 19 per ray-sphere pair,
14 per bounding-sphere test); `roofline.executed` is the arithmetic the kernel
really issues for those pairs (a conservative 15-flop bound per sphere pair, a
23-flop bounding-sphere test per triangle pair, the reference's full
evaluation only for the pairs that survive).  `--full-evaluation` switches the
triangle reject off, so that every pair gets the reference's arithmetic.
`cpu_baseline` times the CPU restatement of the reference's numpy path
(oracle/numpy_path.py, 1 thread) on a bounded sample of the same workload on
this box's host.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "raytracing-one-weekend_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X spec (packed-FP32 issue), 256 CUs at 2.4 GHz

WORKLOADS = {
    # name: (scene builder, width, height, spp, max_bounces, cpu sample (w, h, spp), description)
    "cover": ("cover480", 1280, 720, 100, 10, (320, 180, 8),
              "Book-1 cover scene, 484 spheres (480 random + 3 large + ground), depth-of-field camera"),
    "bunny": ("bunny_diffuse", 1920, 1080, 30, 10, (240, 135, 8),
              "bunny.obj 644 triangles brute force, smooth vertex normals, checker ground sphere"),
    "cloth": ("cloth_bunny", 1920, 1080, 64, 10, (160, 90, 4),
              "textured bunny + UV-mapped picnic cloth (no strength maps: the reference has none)"),
    "demo": ("demo", 320, 180, 10, 5, (320, 180, 10),
             "three-sphere Lambertian/metal/dielectric scene"),
    "cow_teapot": ("cow_teapot", 3840, 2160, 128, 10, (64, 36, 1),
                   "cow.obj + teapot.obj + sphere field; 128 spp per GPU"),
}
# which configuration of BASELINE.json a workload stands for (index into its `configs`, counted from 1)
BASELINE_CONFIG = {"demo": 1, "cover": 2, "bunny": 3, "cloth": 4, "cow_teapot": 5}


def build_scene(kind, aspect):
    from weekend_raytracer import scenes
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == "cover480":
            return scenes.cover_scene(aspect, scenes.sphere_field_records(480))
        return {
            "bunny_diffuse": scenes.bunny_diffuse_scene,
            "cloth_bunny": scenes.cloth_bunny_scene,
            "demo": scenes.numpy_one_weekend_demo_scene,
            "cow_teapot": scenes.cow_teapot_field_scene,
        }[kind](aspect)


def cpu_baseline(kind, sample, max_bounces, seed):
    """The CPU restatement of the reference's numpy path on a bounded sample."""
    from oracle import numpy_path
    from weekend_raytracer import _flatten, main
    w, h, spp = sample
    flat = _flatten.flatten(*build_scene(kind, w / h), main.HORIZON_COLOUR, main.SKY_COLOUR)
    t0 = time.perf_counter()
    r = numpy_path.render(flat, w, h, spp, max_bounces, seed)
    dt = time.perf_counter() - t0
    segments = int(r["counts"][:-1].sum())
    return dict(value=round(segments / dt / 1e6, 5), unit="Mrays/s", cores=1, kind="port",
                sample=f"{w}x{h}, {spp} spp, {max_bounces} bounces of the same scene: {segments} segments in "
                       f"{dt:.1f} s; numpy restatement of the reference path, single thread "
                       f"({os.cpu_count()} host cores present)",
                reference_in_build_container="0.80 Mrays/s demo scene, 0.0205 Mrays/s bunny scene "
                                             "(reference itself, 1 of 8 cores)")


def flop_counts(stats, hierarchy=False):
    """(algorithmic, executed) FP32 flops of the pairs in `stats`.

    algorithmic: what the reference evaluates for these pairs (triangle_group.py:289-395 45 flop,
    sphere_group.py:137-205 19 flop, triangle_group.py:158-166 14 flop).  executed: what the kernel
    issues: spheres meet a 15-flop bound of the discriminant, the full 19 only for candidates (not
    counted); triangles of large groups meet a 23-flop bounding-sphere test and the 45-flop evaluation
    only if they survive it; with the reject off (or in small groups) every pair costs 45 + 10 filter flops.
    With hierarchy=True (--hierarchy) "algorithmic" still counts the pairs the reference would examine,
    so the fraction built on it is a speed-up figure, not a utilisation.
    """
    tri, sph, cull = stats["triangle_pairs"], stats["sphere_pairs"], stats["cull_tests"]
    algorithmic = 45.0 * tri + 19.0 * sph + 14.0 * cull
    evaluated = stats["triangle_evaluations"]
    executed_tri = 23.0 * tri + 45.0 * evaluated if evaluated else 55.0 * tri
    if hierarchy:
        # only the leaf triangles a walk looked at (box tests are not counted): pairs of groups too small
        # for a hierarchy are missing from this figure, it is a lower bound
        executed_tri = 45.0 * evaluated
    return algorithmic, executed_tri + 15.0 * sph + 14.0 * cull


def side_measurement(name, seed, device, steps=3, reject=True, with_cpu=True, hierarchy=False, spp=0):
    """One more workload on this GPU: Mrays/s, roofline fraction and the CPU figure, compactly."""
    import torch
    from weekend_raytracer import main as wr_main
    kind, width, height, full_spp, bounces, cpu_sample, desc = WORKLOADS[name]
    spp = spp or full_spp
    frame = wr_main.Frame(build_scene(kind, width / height), device=device, hierarchy=hierarchy)
    ctx = frame.ctx
    ctx.set_triangle_reject(reject)

    def step():
        accum = ctx.zeros((width, height, 3), torch.float32)
        frame.accumulate(width, height, 0, spp, bounces, seed, accum=accum)
        return frame.image(accum, spp)

    step()
    torch.cuda.synchronize()
    ctx.stats_reset()
    ctx.profile(True)
    ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.profile_read()
    stats = ctx.stats()
    ctx.set_triangle_reject(True)
    ctx.close()
    segments = int(stats["rays_per_bounce"][:bounces].sum())
    flops, executed = flop_counts(stats, hierarchy)
    value = segments / elapsed / 1e6
    out = {"workload": f"{name}: {desc}", "width": width, "height": height, "spp": spp, "max_bounces": bounces,
           "triangle_reject": bool(reject), "triangle_hierarchy": hierarchy,
           "value": round(value, 3), "unit": "Mrays/s", "seconds_per_frame": round(elapsed / steps, 6),
           "kernel_ms_avg": round(kernel_ms / max(1, launches), 4),
           "roofline_frac": round(flops / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
           "executed_frac": round(executed / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
           "triangle_pairs_fully_evaluated": round(stats["triangle_evaluations"] / max(1, stats["triangle_pairs"]), 5)
           if reject or hierarchy else 1.0,
           # lanes that took part in the large groups' loops / lanes those loops ran with
           "lane_efficiency": {"triangle_loops": round(stats["triangle_pairs"] / stats["triangle_lane_slots"], 4)
                               if stats["triangle_lane_slots"] else None,
                               "sphere_loops": round(stats["sphere_pairs"] / stats["sphere_lane_slots"], 4)
                               if stats["sphere_lane_slots"] else None},
           # the bounce kernel's instance with a second and third path per lane in device memory (chosen by the
           # library for scenes with two or more large triangle groups; same results): paths put there per frame
           "path_shelf": {"used": stats["shelved_paths"] > 0, "shelved_paths_per_frame": stats["shelved_paths"] // steps}}
    if out["roofline_frac"] > 1.0:
        # the reference's 45 flop per (ray, triangle) pair against a kernel that settles most pairs with a cheaper
        # conservative test: a speed in the reference's flops, not a share of the VALU peak (that is executed_frac)
        out["roofline_note"] = ("above 1: priced in the reference's dense evaluation of every pair; the kernel's own "
                                "arithmetic is executed_frac of the peak")
    if hierarchy:
        out["lane_efficiency"] = None          # walked groups spend no lane slots in loops: no such figure
        out["roofline_frac"] = None            # not the reference's work: no brute-force roofline figure
        out["speedup_note"] = "box hierarchy: same closest hits bit for bit, fewer pairs examined than the reference does"
    if with_cpu:
        cpu = cpu_baseline(kind, cpu_sample, bounces, seed)
        out.update({"cpu_baseline": cpu, "gpu_over_cpu_port": round(value / cpu["value"], 1),
                    "gpu_over_reference_numpy_in_build_container": round(value / 0.0205, 1)})
    return out


def launch_ranks(n_ranks, argv, job_timeout=540.0, log_dir=None):
    """Start `n_ranks` copies of this script as the ranks of one job (the environment
    torch.distributed.run would give them) from a parent that has made no GPU call.

    Every rank is watched while rank 0's stdout is collected: the first rank that exits non-zero ends
    the job (the others would wait for it in a collective until the backend's timeout), and so does
    `job_timeout` seconds without all ranks done.  Rank 0's stdout (the one JSON line) is passed on only
    when every rank succeeded; each rank's stderr goes to `bench_rank<r>.err` under `log_dir` (default:
    BENCH_LOG_DIR or the system's temporary directory) and the failing rank's last lines are shown.
    -> the first non-zero exit status of a rank (124 for the time limit), else 0."""
    import socket
    import subprocess
    import tempfile
    import threading
    log_dir = log_dir or os.environ.get("BENCH_LOG_DIR") or tempfile.gettempdir()
    os.makedirs(log_dir, exist_ok=True)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ranks, logs = [], []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL between processes needs here
        log_path = os.path.join(log_dir, f"bench_rank{rank}.err")
        log = open(log_path, "wb")
        logs.append((log_path, log))
        ranks.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=log))
    collected = []

    def collect():                                   # rank 0 prints the one JSON line
        for line in ranks[0].stdout:
            collected.append(line.decode())

    reader = threading.Thread(target=collect, daemon=True)
    reader.start()
    status, failed = 0, None
    deadline = time.monotonic() + job_timeout
    try:
        while status == 0:
            codes = [proc.poll() for proc in ranks]
            for rank, rc in enumerate(codes):
                if rc not in (None, 0):
                    status, failed = rc, rank
                    break
            if status == 0 and all(rc == 0 for rc in codes):
                break
            if status == 0 and time.monotonic() > deadline:
                status = 124
                sys.stderr.write(f"bench.py launcher: the job did not finish within {job_timeout:.0f} s\n")
            if status == 0:
                time.sleep(0.05)
    finally:
        for proc in ranks:                           # a failed rank leaves the others waiting in a collective
            if proc.poll() is None:
                proc.terminate()
        for proc in ranks:
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
        reader.join(timeout=10)
        for _, log in logs:
            log.close()
    if status == 0:
        sys.stdout.write("".join(collected))
        sys.stdout.flush()
        return 0
    sys.stderr.write("".join(collected))             # whatever rank 0 said is not a result of a failed job
    for rank, (log_path, _) in enumerate(logs):
        if failed is None or rank == failed:
            with open(log_path, "rb") as fh:
                tail = fh.read()[-3000:].decode(errors="replace")
            sys.stderr.write(f"--- rank {rank} (exit status {ranks[rank].returncode}), {log_path}:\n{tail}\n")
    return status if status > 0 else 1               # (a rank killed by a signal has a negative status)


def launch_check(world, rank, backend):
    """--launch-check: the ranks meet, count themselves over the backend and leave (no rendering);
    what the CPU test of the launcher runs."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend)
    hook = os.environ.get("BENCH_FAIL_RANK")                # test hook of the launcher's CPU tests
    if hook and world > 1:
        if hook == str(rank):                               # this rank dies once the ranks have met ...
            os._exit(3)
        if hook == "hang" and rank == world - 1:            # ... or never comes to the collective
            time.sleep(3600)
        dist.barrier()                                      # the others wait for it in a collective
    seen = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"launch_check": True, "ranks_seen": int(seen.item()), "n_gpus": world}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cover", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--full-evaluation", action="store_true",
                    help="no bounding-sphere reject: the reference's full evaluation for every ray-triangle pair")
    ap.add_argument("--hierarchy", action="store_true",
                    help="optional mode, not the reference's algorithm: large triangle groups are walked through a box "
                         "hierarchy; same image bit for bit, and a figure to be read apart from the brute-force one")
    ap.add_argument("--hierarchy-from", type=int, default=0, metavar="N",
                    help="like --hierarchy, for groups of N or more triangles (default threshold: 1024)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = the workload's spp on every GPU; strong = the workload's spp split over the GPUs")
    ap.add_argument("--job-timeout", type=float, default=540.0,
                    help="launcher only (python bench.py --gpus N from a plain shell): seconds after which the job is ended")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, count them over the backend and stop (tests the launcher; no rendering)")
    args = ap.parse_args()
    if args.hierarchy_from:
        args.hierarchy = args.hierarchy_from

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], job_timeout=args.job_timeout))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (before anything initialises the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 code path on a
    # one-GPU box); the driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if args.launch_check:
        return launch_check(world, rank, backend)

    import torch
    import torch.distributed as dist
    from weekend_raytracer import distributed
    from weekend_raytracer import main as wr_main

    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    kind, width, height, spp, bounces, cpu_sample, desc = WORKLOADS[args.workload]
    width, height, spp = args.width or width, args.height or height, args.spp or spp
    frame = wr_main.Frame(build_scene(kind, width / height), device=local_rank, hierarchy=args.hierarchy)
    ctx = frame.ctx
    ctx.set_triangle_reject(not args.full_evaluation)
    # weak: every rank renders `spp` samples; strong: the `spp` samples are split over the ranks
    spp_total = spp * world if args.scaling == "weak" else spp
    if spp_total < world:
        raise SystemExit(f"--scaling strong: {spp_total} samples per pixel cannot be split over {world} ranks")

    def all_reduce_sum(accum):
        if world > 1:
            distributed.all_reduce_sum(accum)       # RCCL in place; a gloo rehearsal goes through the host

    def render_samples(s_begin, s_end):
        accum = ctx.zeros((width, height, 3), torch.float32)
        frame.accumulate(width, height, s_begin, s_end, bounces, args.seed, accum=accum)
        return accum

    def step():
        # the package's own multi-GPU routine: this rank's contiguous share of the global sample
        # indices, one all-reduce of the float32 accumulation images
        accum = distributed.accumulate_sharded(render_samples, spp_total, rank, world, all_reduce_sum)
        return frame.image(accum, spp_total)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.stats_reset()
    ctx.profile(True)
    ctx.profile_read()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image = step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    kernel_ms, launches = ctx.profile_read()
    stats = ctx.stats()

    segments = int(stats["rays_per_bounce"][:bounces].sum())
    flops, executed_flops = flop_counts(stats, args.hierarchy)
    totals = torch.tensor([float(segments), elapsed, 1.0], dtype=torch.float64,
                          device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        sums = totals.clone()                       # segments and a count of the ranks the collective saw
        t_max = totals[1:2].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        segments_all, elapsed_max, ranks_seen = float(sums[0].item()), float(t_max.item()), int(sums[2].item())
    else:
        segments_all, elapsed_max, ranks_seen = float(segments), elapsed, 1
    s_mine = distributed.shard_samples(spp_total, rank, world)

    if rank == 0:
        # The reference's refraction is not renormalised (materials.py:395-398); once in ~3e7 glass paths a
        # direction blows up and a sample's colour leaves [0, 1] (the CPU oracle reproduces those paths bit
        # for bit), so sqrt(mean) can be NaN in a handful of pixels, exactly as numpy.sqrt gives the reference.
        bad_pixels = int((~torch.isfinite(image).all(dim=2)).sum().item())
        # what was rendered, so that lines from runs with different N can be told to show the same picture:
        # the image's mean over its finite pixels (equal over N up to the float32 order of the ranks' sums)
        # and a CRC of its 8-bit form (equal unless a sum lands within that order of a quantisation step)
        import zlib
        finite = torch.where(torch.isfinite(image), image, torch.zeros_like(image))
        image_check = {"mean": round(float(finite.double().mean().item()), 7),
                       "bytes_crc32": zlib.crc32(ctx.image_bytes(image).cpu().numpy().tobytes())}
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        # mean wavefront lifetime in shader clock cycles over the average kernel duration (the wavefronts are
        # resident for the whole launch; the chip's clock domains differ by a few per cent, this is their mean)
        clock_mhz = round(float(stats["clock_ticks"][0]) / max(1, stats["wavefronts"]) / (kernel_ms / max(1, launches)) / 1e3, 1) \
            if kernel_ms > 0 else None
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                rec = json.load(fh).get(args.workload)
            if rec and rec.get("width") == width and rec.get("height") == height and rec.get("spp") == spp:
                traffic = rec["hbm_bytes_per_launch"]
        out = {
            "metric": "Mrays/sec (all bounce segments)",
            "value": round(segments_all / elapsed_max / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "seconds_per_frame": round(elapsed_max / args.steps, 6),
            "higher_is_better": True,
            "scaling": args.scaling,
            "ranks_seen": ranks_seen,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            # the configuration of BASELINE.json this line's scene belongs to, and whether the line is that
            # configuration as stated (size, samples, GPU count: config 5 is 8 GPUs x 128 spp, the others one
            # GPU) or its scene at another size, sample count or GPU count
            "baseline_config": BASELINE_CONFIG[args.workload],
            "baseline_config_exact": not (args.spp or args.width or args.height) and
            (world == (8 if args.workload == "cow_teapot" else 1)) and (world == 1 or args.scaling == "weak"),
            "image_check": image_check,
            "config": {
                "workload": f"{args.workload}: {desc}", "width": width, "height": height,
                "spp_per_gpu": s_mine[1] - s_mine[0], "spp_total": spp_total, "max_bounces": bounces,
                "seed": args.seed, "backend": "rccl" if backend == "nccl" else backend,
                "segments_per_frame_per_gpu": segments // max(1, args.steps),
                "parallelism": f"samples sharded over {world} GPU(s), RCCL all-reduce of the fp32 image"
                               if world > 1 else "single GPU",
                "non_finite_pixels": bad_pixels,
                "triangle_hierarchy": args.hierarchy,
            },
            "roofline": {
                "bound": "valu",
                "achieved": round(achieved, 3),
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4),
                # achieved / frac price the reference's dense evaluation of the same pairs (a speed figure in the
                # reference's own flops); the share of the VALU peak the kernel's own arithmetic uses is frac_executed
                "achieved_basis": "algorithmic flops of the reference's evaluation of these pairs",
                "frac_executed": round(executed_flops / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)
                if kernel_ms > 0 else 0.0,
                "traffic": traffic,
                # the shader clock the kernel ran at (the peak assumes 2400 MHz) and the fraction priced at that clock
                "shader_clock_mhz": clock_mhz,
                "frac_at_clock": round(achieved / FP32_VECTOR_PEAK_TFLOPS * 2400.0 / clock_mhz, 4) if clock_mhz else None,
                "kernel": "k_render",
                "kernel_ms_avg": round(kernel_ms / max(1, launches), 4),
                "launches": launches,
                "algorithmic_flop_per_launch": flops / max(1, launches),
                "executed": {"tflops": round(executed_flops / (kernel_ms * 1e-3) / 1e12, 3) if kernel_ms > 0 else 0.0,
                             "frac": round(executed_flops / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)
                             if kernel_ms > 0 else 0.0,
                             "flop_per_launch": executed_flops / max(1, launches),
                             "triangle_reject": not args.full_evaluation,
                             "triangle_pairs_fully_evaluated": stats["triangle_evaluations"] // max(1, launches)},
                "pairs_per_launch": {"triangle": stats["triangle_pairs"] // max(1, launches),
                                     "sphere": stats["sphere_pairs"] // max(1, launches),
                                     "cull": stats["cull_tests"] // max(1, launches)},
                # the same launch against the HBM roofline: algorithmic bytes (48 B of ray state per
                # segment as the reference moves it, 12 B accumulated per primary ray); the kernel itself keeps
                # ray state in registers and writes 16 B per path
                "hbm_view": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                             "achieved": round(float(48.0 * segments + 12.0 * float(stats["rays_per_bounce"][0])) /
                                               (kernel_ms * 1e-3) / 1e9, 2)
                             if kernel_ms > 0 else 0.0},
                "lane_efficiency": {
                    # (with the hierarchy, walked groups spend no lane slots in loops: no figure then)
                    "triangle_loops": round(stats["triangle_pairs"] / stats["triangle_lane_slots"], 4)
                    if stats["triangle_lane_slots"] and not args.hierarchy else None,
                    "sphere_loops": round(stats["sphere_pairs"] / stats["sphere_lane_slots"], 4)
                    if stats["sphere_lane_slots"] and not args.hierarchy else None},
                "note": "FP32 vector (VALU) roofline: this path is not a dense contraction, MFMA is not used; "
                        "achieved = 45/ray-triangle pair + 19/ray-sphere pair + 14/cull test as the reference "
                        "evaluates them, over the kernel time; executed = the cheaper conservative tests the "
                        "kernel issues for most pairs plus the full evaluation of the survivors (same results "
                        "bit for bit)",
            },
        }
        if not args.no_cpu and world == 1:           # the host leg runs on rank 0 of a one-GPU job only
            out["cpu_baseline"] = cpu_baseline(kind, cpu_sample, bounces, args.seed)
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if world == 1 and args.workload == "cover" and not args.no_cpu and not (args.spp or args.width or args.height):
            # the mesh configuration the acceptance target is stated on, measured the same way (single GPU)
            out["also"] = {"bunny": side_measurement("bunny", args.seed, local_rank),
                           "bunny_full_evaluation": side_measurement("bunny", args.seed, local_rank, reject=False,
                                                                     with_cpu=False),
                           # optional mode, reported apart: not the reference's brute-force work
                           "bunny_hierarchy": side_measurement("bunny", args.seed, local_rank, with_cpu=False,
                                                               hierarchy=64),   # below the default threshold of 1024
                           # optional mode on the sphere scene (484 spheres: about where walking a sphere group
                           # starts to pay), reported apart
                           "cover_hierarchy": side_measurement("cover", args.seed, local_rank, with_cpu=False,
                                                               hierarchy=64),
                           # textured bunny + UV-mapped cloth at its stated size
                           "cloth": side_measurement("cloth", args.seed, local_rank, steps=2, with_cpu=False),
                           # cow + teapot + sphere field at its stated per-GPU size (3840x2160, 128 spp)
                           "cow_teapot_128spp": side_measurement("cow_teapot", args.seed, local_rank, steps=1,
                                                                 with_cpu=False),
                           "cow_teapot_128spp_hierarchy": side_measurement("cow_teapot", args.seed, local_rank,
                                                                           steps=1, with_cpu=False, hierarchy=True)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
