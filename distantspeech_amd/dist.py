"""Multi-GPU plumbing: one process per GPU, utterances sharded by rank, no collective on the data
path (utterances never interact).  The only exchange is the final throughput
reduction (SUM of frames, MAX of elapsed) over torch.distributed — RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests."""
import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total, rank, world):
    """contiguous shard [lo, hi) of `total` utterances for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend=None):
    """Select this rank's GPU, then init torch.distributed from the environment when WORLD_SIZE > 1.
    Returns (rank, local_rank, world).  Env overrides for single-GPU testing: DS_DIST_BACKEND (e.g. gloo),
    DS_FORCE_DEVICE (ordinal every rank should use), DS_DIST_FORCE=1 (form the process group even at WORLD_SIZE = 1, so that
    the collective path — RCCL with backend "nccl": device tensors, barrier(device_ids) — runs on a one-GPU box)."""
    rank, local_rank, world = env_world()
    dev = int(os.environ.get("DS_FORCE_DEVICE", local_rank))
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(dev)          # before the process group exists: RCCL binds to the current device
    except ImportError:
        torch = None
    if world > 1 or os.environ.get("DS_DIST_FORCE") == "1":
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = os.environ.get("DS_DIST_BACKEND") or ("nccl" if torch is not None and torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, dev, world


def barrier():
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def collective_name():
    """backend of the process group the reductions run over ("nccl" = RCCL, "gloo"), or "none" (single process, no group)"""
    dist = _group()
    return dist.get_backend() if dist is not None else "none"


def _group():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def reduce_throughput(frames, elapsed_s, device=None):
    """(total frames over ranks, max elapsed over ranks, ranks that contributed): the one collective of the job."""
    dist = _group()
    if dist is None:
        return int(frames), float(elapsed_s), 1
    import torch
    dev = device if dist.get_backend() == "nccl" else None
    f = torch.tensor([float(frames), 1.0], dtype=torch.float64, device=dev)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(f[0].item())), float(t.item()), int(round(f[1].item()))


def reduce_max(value, device=None):
    """MAX of a scalar over ranks (set-up decisions that every rank must take alike, and device timings)."""
    dist = _group()
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def finalize():
    dist = _group()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
