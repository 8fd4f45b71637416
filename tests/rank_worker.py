"""
One rank of the two-rank GPU test of the sample-sharded render (tests/test_gpu_distributed.py starts
two of these; they share the GPU and meet over gloo).  It runs the product routine,
distributed.accumulate_sharded over Frame.accumulate, and saves this rank's own shard sum and the
all-reduced accumulation image.

    python tests/rank_worker.py <rank> <world> <port> <out_dir> <fixture> <samples>
"""
import contextlib
import io
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "raytracing-one-weekend_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, out_dir, fixture, samples = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                    sys.argv[5], int(sys.argv[6]))
    import numpy as np
    import torch
    import torch.distributed as dist
    from tests import helpers
    from weekend_raytracer import distributed, main as wr_main, scenes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = helpers.golden(fixture)
    w, h, _, b, _ = (int(v) for v in f["size"])
    seed = int(f["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        frame = wr_main.Frame(scenes.numpy_one_weekend_demo_scene(w / h), device=0)
    own = {}

    def render_samples(s_begin, s_end):
        accum = frame.ctx.zeros((w, h, 3), torch.float32)
        frame.accumulate(w, h, s_begin, s_end, b, seed, accum=accum)
        own["span"], own["sum"] = (s_begin, s_end), accum.clone()
        return accum

    accum = distributed.accumulate_sharded(render_samples, samples, rank, world, distributed.all_reduce_sum)
    image = frame.image(accum, samples)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), span=np.array(own["span"]), own=own["sum"].cpu().numpy(),
             accum=accum.cpu().numpy(), image=image.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    frame.ctx.close()


if __name__ == "__main__":
    main()
