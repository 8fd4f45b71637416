"""
Multi-GPU rendering: one process per GPU, samples per pixel sharded across
ranks, float32 accumulation images summed with one all-reduce (RCCL over xGMI
when the backend is "nccl").  The reference is single-process; its sample
loop is the axis `samples` of main.py:96-104,219, which is what gets split.

Random streams are keyed by the global sample index, so the image does not
depend on how samples are partitioned (up to float32 summation order).
"""


def shard_samples(total_samples, rank, world_size):
    """Contiguous, balanced split of sample indices [0, total) -> [begin, end)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside the world")
    base, extra = divmod(int(total_samples), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def accumulate_sharded(render_samples, total_samples, rank, world_size, all_reduce_sum):
    """render_samples(s_begin, s_end) -> this rank's accumulation image (sum over
    its samples; None when the shard is empty); all_reduce_sum(tensor) sums it
    over ranks in place.  -> the accumulation image of all samples, on every rank."""
    s_begin, s_end = shard_samples(total_samples, rank, world_size)
    accum = render_samples(s_begin, s_end)
    all_reduce_sum(accum)
    return accum


def all_reduce_sum(accum, group=None):
    """Sum a CUDA accumulation image over the ranks of the process group, in place.  With the
    "nccl" backend (RCCL on ROCm) the tensor is reduced where it lies, over xGMI; a host-side
    backend (gloo: several ranks sharing one GPU, rehearsals without RCCL) goes through host memory."""
    import torch.distributed as dist
    if dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    else:
        host = accum.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        accum.copy_(host)


def render_distributed(frame, width, height, total_samples, max_bounces, seed, group=None):
    """Render with every rank of the default (or given) torch.distributed
    process group; -> (image CUDA tensor on every rank, accum)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def render_samples(s_begin, s_end):
        accum = frame.ctx.zeros((width, height, 3), torch.float32)
        if s_end > s_begin:
            frame.accumulate(width, height, s_begin, s_end, max_bounces, seed, accum=accum)
        return accum

    accum = accumulate_sharded(render_samples, total_samples, rank, world, lambda t: all_reduce_sum(t, group))
    return frame.image(accum, total_samples), accum
