"""Test infrastructure only: a GPU-less stand-in for bench.py's backend so that the rank plumbing of `bench.py --gpus N`
(self-launch of N child ranks, gloo rendezvous on 127.0.0.1, barrier / MAX-over-ranks bracketing, the one JSON line, the
rank-count check) runs in the CPU container.  Selected with DS_BENCH_BACKEND=tests.bench_stub:StubBackend; it computes nothing
and the line it produces says so in `data`."""
import os
import time


class StubWorkload:
    def __init__(self, w, B, T, K, W):
        self.steps = 0
        self.t0 = 0.0
        self.step_s = float(os.environ.get("DS_BENCH_STUB_STEP_S", "2e-5"))
        self.pending = 0.0

    def run(self, first_step, n):
        self.steps += n
        self.pending += n * self.step_s           # "enqueued" work, waited for at the next synchronisation point

    def _drain(self):
        if self.pending > 0:
            time.sleep(self.pending)
            self.pending = 0.0

    def sync(self):
        self._drain()

    def timing_begin(self):
        self._drain()
        self.t0 = time.perf_counter()

    def timing_end(self):
        self._drain()
        return (time.perf_counter() - self.t0) * 1e3

    def check(self, first_step):
        pass

    def close(self):
        log = os.environ.get("DS_BENCH_STUB_LOG")
        if log:
            with open("%s.%s" % (log, os.environ.get("RANK", "0")), "a") as fh:
                fh.write("%d\n" % self.steps)


class StubBackend:
    name = "stub (no GPU: rank-plumbing test, nothing is computed)"
    dist_backend = "gloo"

    def __init__(self, local_rank, world):
        self.local_rank = local_rank
        self.device = None
        self.made = []

    def device_sync(self):
        for wl in self.made:
            wl._drain()

    def make(self, w, B, T, K, W, seed, graph, split=None):
        wl = StubWorkload(w, B, T, K, W)
        self.made = [wl]
        return wl
