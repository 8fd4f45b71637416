"""
The multi-GPU path on CPU: two gloo ranks shard the samples, render their
shard (the CPU oracle stands in for the GPU renderer), all-reduce the
accumulation images and must reproduce the single-process image.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_path
from tests import helpers
from weekend_raytracer import distributed


def test_shard_samples_partition():
    for total in (0, 1, 7, 100, 1024):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_samples(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = helpers.golden("frame_demo")
    scene = helpers.golden_scene(f)
    w, h, s, b, _ = (int(v) for v in f["size"])
    w, h = 24, 14                                   # keep the CPU suite short
    seed = int(f["seed"])

    def render_samples(s_begin, s_end):
        r = numpy_path.render(scene, w, h, s_end - s_begin, b, seed, s_begin=s_begin)
        return torch.from_numpy(np.ascontiguousarray(r["sample_sum"]))

    accum = distributed.accumulate_sharded(render_samples, s, rank, world, lambda t: dist.all_reduce(t))
    np.save(os.path.join(out_dir, f"accum{rank}.npy"), accum.numpy())
    dist.destroy_process_group()


def test_two_rank_sample_sharding_matches_single_process(tmp_path):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = (np.load(tmp_path / f"accum{r}.npy") for r in range(2))
    assert np.array_equal(a0, a1)                   # every rank holds the full sum
    f = helpers.golden("frame_demo")
    scene = helpers.golden_scene(f)
    s, b = int(f["size"][2]), int(f["size"][3])
    whole = numpy_path.render(scene, 24, 14, s, b, int(f["seed"]))
    # identical samples, float32 additions associated differently
    assert np.allclose(a0, whole["sample_sum"], rtol=1e-6, atol=1e-6)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell: the parent starts two ranks of itself (no
    torch.distributed.run), they meet over gloo, and rank 0's JSON line comes back through the parent."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(helpers.REPO, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch_check": True, "ranks_seen": 2, "n_gpus": 2}


def test_bench_launcher_reports_a_failed_rank():
    import subprocess
    import sys
    env = dict(os.environ, BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # strong scaling cannot split 1 sample over 2 ranks ... but without a GPU the ranks already fail at
    # the device: either way the parent must come back non-zero and print no JSON line
    run = subprocess.run([sys.executable, os.path.join(helpers.REPO, "bench.py"), "--gpus", "2", "--workload", "demo",
                          "--spp", "1", "--scaling", "strong", "--no-cpu", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_ends_the_job_when_one_rank_dies(tmp_path):
    """Rank 1 exits with status 3 after init_process_group while rank 0 waits in a barrier: the parent
    must notice at once (not after the backend's timeout), end rank 0, return non-zero, print no JSON
    line and leave rank 1's stderr log where it says."""
    import subprocess
    import sys
    import time
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_FAIL_RANK="1", BENCH_LOG_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    began = time.monotonic()
    run = subprocess.run([sys.executable, os.path.join(helpers.REPO, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert time.monotonic() - began < 30.0
    assert run.returncode == 3, (run.returncode, run.stderr[-2000:])
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert (tmp_path / "bench_rank1.err").exists() and "rank 1 (exit status 3)" in run.stderr


def test_bench_launcher_time_limit(tmp_path):
    """A job that does not finish is ended at --job-timeout with status 124."""
    import subprocess
    import sys
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_FAIL_RANK="hang", BENCH_LOG_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(helpers.REPO, "bench.py"), "--gpus", "2", "--launch-check",
                          "--job-timeout", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert run.returncode == 124 and "did not finish within 8 s" in run.stderr
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
