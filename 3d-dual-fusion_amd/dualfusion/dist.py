"""One-process-per-GPU data parallelism for the hot path.

Frames are independent through the whole path, so ranks process disjoint frames and the data path
has NO collective.  The only cross-rank traffic the reference has on this path is the reduction of
the loss/log scalars (`reduce_dict`, CP/det3d/torchie/trainer/utils.py:157-183; `all_reduce`,
VR/pcdet/utils/commu_utils.py:148-162) and the timing/barrier protocol of the benchmark.  Backend
"nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (rank, local_rank, world_size); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, init_method="env://")
    return rank, local, world


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def frame_shard(num_frames, rank, world):
    """DistributedSampler-style frame assignment without padding: frame i goes to rank i % world."""
    return list(range(rank, num_frames, world))


def barrier(device=None):
    if is_dist():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value, device="cpu"):
    """Elapsed time of the slowest rank (the driver contract: barrier, time, MAX over ranks)."""
    if not is_dist():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_dict(input_dict, average=True):
    """CP/det3d/torchie/trainer/utils.py:157-183: stack the (sorted-key) scalar tensors, reduce to
    rank 0, average there.  Other ranks get the un-divided partial result back, like the reference."""
    if not is_dist():
        return input_dict
    world = dist.get_world_size()
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return {k: v for k, v in zip(names, values)}


def all_reduce_value(data, op="sum", average=False):
    """VR/pcdet/utils/commu_utils.py:148-162."""
    if not is_dist():
        return data
    ops = {"SUM": dist.ReduceOp.SUM, "MAX": dist.ReduceOp.MAX, "MIN": dist.ReduceOp.MIN,
           "PRODUCT": dist.ReduceOp.PRODUCT}
    out = data.clone()
    dist.all_reduce(out, op=ops[op.upper()])
    if average:
        assert op.upper() == "SUM"
        return out / dist.get_world_size()
    return out
