"""Multi-process path on CPU (gloo, world_size 2): shard arithmetic and the final throughput reduction
(the only collective of the job; on the GPU box the same code runs over RCCL)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from distantspeech_amd.dist import shard_range
    for total in (1, 7, 1024, 1025, 16384):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from distantspeech_amd import dist as d
    r, lr, w = d.init(backend="gloo")
    lo, hi = d.shard_range(1025, r, w)
    d.barrier()
    frames, t, ranks = d.reduce_throughput((hi - lo) * 10, 1.0 + r)      # rank 1 is "slower"
    assert ranks == w and d.reduce_max(3.0 + r) == 4.0
    assert d.gather_ints([7, -(2 ** 62) - r]) == [[7, -(2 ** 62)], [7, -(2 ** 62) - 1]]      # bench.py --verify: every rank sees every rank's words
    q.put((r, lo, hi, frames, t))
    d.finalize()


def test_numa_pinning_never_raises_and_keeps_a_core():
    from distantspeech_amd.dist import gather_ints, pin_to_gpu_numa_node
    before = os.sched_getaffinity(0)
    try:
        for dev in (0, 7, 99):
            what = pin_to_gpu_numa_node(dev)
            assert isinstance(what, str) and len(os.sched_getaffinity(0)) >= 1
    finally:
        os.sched_setaffinity(0, before)
    assert gather_ints([1, 2]) == [[1, 2]]                                # no process group: this rank alone


def test_gloo_world2_reduce():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert [r[1:3] for r in res] == [(0, 513), (513, 1025)]
    assert all(r[3] == 10250 and abs(r[4] - 2.0) < 1e-9 for r in res)     # SUM of frames, MAX of elapsed
