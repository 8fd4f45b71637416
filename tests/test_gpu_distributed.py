"""
The multi-GPU path on the GPU: distributed.render_distributed under torch.distributed with the
RCCL backend (one rank: a one-GPU box has no second device for RCCL), and bench.py started from a
plain shell with two ranks that share the GPU and reduce over gloo.
"""
import contextlib
import io
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_render_distributed_over_rccl_one_rank():
    import torch
    import torch.distributed as dist
    from weekend_raytracer import distributed, main, scenes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        f = helpers.golden("frame_demo")
        w, h, s, b, _ = (int(v) for v in f["size"])
        with contextlib.redirect_stdout(io.StringIO()):
            frame = main.Frame(scenes.numpy_one_weekend_demo_scene(w / h))
        image, accum = distributed.render_distributed(frame, w, h, s, b, int(f["seed"]))
        torch.cuda.synchronize()
        assert np.array_equal(accum.cpu().numpy(), f["sample_sum"])
        assert np.array_equal(image.cpu().numpy(), f["image"])
        frame.ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_from_a_plain_shell(scaling):
    env = dict(os.environ, BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(helpers.REPO, "bench.py"), "--gpus", "2", "--workload", "demo",
                          "--no-cpu", "--steps", "2", "--warmup", "1", "--scaling", scaling],
                         env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == scaling
    assert out["config"]["spp_total"] == (20 if scaling == "weak" else 10)
    assert out["config"]["spp_per_gpu"] == (10 if scaling == "weak" else 5)
    assert out["value"] > 0 and out["roofline"]["kernel_ms_avg"] > 0
    assert out["baseline_config"] == 1 and out["baseline_config_exact"] is False
    assert out["image_check"]["bytes_crc32"] > 0 and 0.0 < out["image_check"]["mean"] < 1.0


def test_two_rank_image_equals_the_one_rank_image(tmp_path):
    """Two ranks (sharing the GPU, gloo) run the product routine - distributed.accumulate_sharded over
    Frame.accumulate, what bench.py's step runs - on the demo frame, its samples split in halves: each rank's
    own sum is bit for bit the single-process render of its span of sample indices (the sample axis of
    reference main.py:96-104), the all-reduced image is on both ranks and equals the one-rank image up
    to the float32 order of the two partial sums, and its sqrt(mean) (main.py:219-220) likewise."""
    import torch
    from weekend_raytracer import distributed, main, scenes
    f = helpers.golden("frame_demo")
    w, h, s, b, _ = (int(v) for v in f["size"])
    port = free_port()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    worker = os.path.join(helpers.REPO, "tests", "rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path), "frame_demo", str(s)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    with contextlib.redirect_stdout(io.StringIO()):
        frame = main.Frame(scenes.numpy_one_weekend_demo_scene(w / h))
    seed = int(f["seed"])
    for r, (lo, hi) in enumerate(distributed.shard_samples(s, r, 2) for r in range(2)):
        assert hi > lo
        assert tuple(ranks[r]["span"]) == (lo, hi)
        alone = frame.accumulate(w, h, lo, hi, b, seed)
        torch.cuda.synchronize()
        assert np.array_equal(ranks[r]["own"], alone.cpu().numpy()), r          # bit for bit
    assert np.array_equal(ranks[0]["accum"], ranks[1]["accum"])                   # every rank holds the sum
    assert np.array_equal(ranks[0]["accum"], ranks[0]["own"] + ranks[1]["own"])
    # one rank adds sample after sample; two ranks add two partial sums: float32 order only
    assert np.allclose(ranks[0]["accum"], f["sample_sum"], rtol=1e-6, atol=0.0)
    assert np.allclose(ranks[0]["image"], f["image"], rtol=1e-6, atol=0.0)
    frame.ctx.close()
