"""One-process-per-GPU data parallelism for the hot path.

Frames are independent through the whole path, so ranks process disjoint frames and the data path
has NO collective.  The only cross-rank traffic the reference has on this path is the reduction of
the loss/log scalars (`reduce_dict`, CP/det3d/torchie/trainer/utils.py:157-183; `all_reduce`,
VR/pcdet/utils/commu_utils.py:148-162) and the timing/barrier protocol of the benchmark.  Backend
"nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(nprocs, script, argv, master_port=None, env=None, timeout=None):
    """Run `script argv...` as `nprocs` ranks of ONE node, one process per GPU, the way the reference launches its
    trainers (`python -m torch.distributed.launch --nproc_per_node=N tools/train.py`, CP/docs/GETTING_START.md;
    the worker side is CP/det3d/torchie/apis/train.py:289-295): torch.distributed.run with a 127.0.0.1 rendezvous
    exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* to every rank, which `init_from_env` reads.  Returns the exit
    code of the launcher (0 only when every rank exited cleanly)."""
    port = int(master_port) if master_port else free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nprocs)),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, int(nprocs)))))
    return subprocess.call(cmd, env=e, timeout=timeout)


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
    """CP/det3d/torchie/trainer/utils.py:157-183: the values of the (sorted-key) dict go through ONE reduce to
    rank 0, which averages them; the other ranks get the un-divided partial result back, like the reference.
    The reference stacks scalars; here the values are flattened into one buffer, so per-task vectors (`CenterHead`
    returns one value per task) and entries of different shapes travel in the same single collective."""
    if not is_dist():
        return input_dict
    world = dist.get_world_size()
    with torch.no_grad():
        names = sorted(input_dict.keys())
        parts = [input_dict[k].reshape(-1) for k in names]
        values = torch.cat(parts) if len(parts) > 1 else parts[0].clone()
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        out, pos = {}, 0
        for k, p in zip(names, parts):
            out[k] = values[pos:pos + p.numel()].view(input_dict[k].shape)
            pos += p.numel()
        return out


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
