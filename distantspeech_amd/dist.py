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


def gather_ints(values, device=None):
    """all ranks' lists of int64 values, as a list (one list per rank, in rank order) on every rank — the cross-rank comparison of `bench.py
    --verify` (every rank computes the same utterances; their checksums must agree bit for bit)."""
    vals = [int(v) for v in values]
    dist = _group()
    if dist is None:
        return [vals]
    import torch
    dev = device if dist.get_backend() == "nccl" else None
    mine = torch.tensor(vals, dtype=torch.int64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[int(v) for v in t.tolist()] for t in out]


def pin_to_gpu_numa_node(device_ordinal):
    """Restrict this process to the CPU cores of the NUMA node its GPU hangs off (sysfs: /sys/class/drm/card*/device/numa_node and the node's
    cpulist), so that a rank's launch thread and its pinned staging buffers sit next to its GPU on a multi-socket node.  Call it BEFORE the
    first GPU call of the process.  Returns a short description of what was done ("numa node 1: 48 cores", "no numa information", ...); never
    raises — affinity is an optimisation, not a requirement.  DS_NO_AFFINITY=1 turns it off."""
    if os.environ.get("DS_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return "off"
    try:
        import glob
        import re
        # render nodes of AMD GPUs in PCI order = HIP's device order on a single-vendor node
        cards = []
        for path in glob.glob("/sys/class/drm/card[0-9]*"):
            if not re.fullmatch(r"card\d+", os.path.basename(path)):
                continue
            try:
                vendor = open(os.path.join(path, "device", "vendor")).read().strip()
                if vendor != "0x1002" or not os.path.exists(os.path.join(path, "device", "mem_info_vram_total")):
                    continue
                cards.append((os.path.basename(os.path.realpath(os.path.join(path, "device"))), path))
            except OSError:
                continue
        cards.sort()
        # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber the devices the process sees
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            try:
                order = [int(v) for v in vis.split(",") if v.strip() != ""]
                cards = [cards[i] for i in order if 0 <= i < len(cards)]
            except ValueError:
                pass
        if not (0 <= int(device_ordinal) < len(cards)):
            return "no numa information"
        node = int(open(os.path.join(cards[int(device_ordinal)][1], "device", "numa_node")).read().strip())
        if node < 0:
            return "no numa information"
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return "no numa information"
        os.sched_setaffinity(0, allowed)
        return "numa node %d: %d cores" % (node, len(allowed))
    except (OSError, ValueError):
        return "no numa information"


def finalize():
    dist = _group()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
