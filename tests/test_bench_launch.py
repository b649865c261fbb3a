"""`bench.py --gpus N` starts its N ranks itself: end-to-end under gloo with the stub backend of
tests/bench_stub.py — the one JSON line carries n_gpus = N and the frames of every rank; a rank-count mismatch exits non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=300):
    env = dict(os.environ, DS_BENCH_BACKEND="tests.bench_stub:StubBackend", PYTHONPATH=ROOT, DS_BENCH_DETAIL=os.devnull)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DS_DIST_FORCE", "DS_DIST_BACKEND"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks(tmp_path):
    log = str(tmp_path / "steps")
    detail = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--min-region-ms", "30"], {"DS_BENCH_STUB_LOG": log, "DS_BENCH_DETAIL": detail})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 only, ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["data"].startswith("stub")
    R = out["rounds"]
    # (the stub's steps are sleeps: a loaded host stretches the probe round the round count is sized from, so the region's LENGTH is not
    # asserted against --min-region-ms here — only that whole rounds were timed and that the figures below are consistent with it)
    assert R >= 1 and out["timed_steps"] == 20 * R and out["region_ms"] > 0
    # whole-job value: both ranks' frames over the slowest rank's region
    frames = 2 * 1024 * 20 * R
    assert abs(out["value"] - frames / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3
    assert abs(out["ms_per_step"] - out["region_ms"] / (20 * R)) < 1e-3
    assert out["collective"] == "gloo"
    assert out["verify"] == "ok" and out["config"]["verify"].startswith("2 rank(s)")       # on by default with more than one rank
    assert set(out["other_configs"]) >= {"mvdr_pf", "mvdr_pf_10s_chunks", "cfg3", "cfg4", "cfg5", "cfg2_10s_chunks", "cfg3_10s_chunks", "cfg4_10s_chunks", "cfg5_10s_chunks"}
    assert all(set(v) >= {"value", "ms_per_step", "bound", "frac"} for v in out["other_configs"].values())
    assert out["roofline"]["bound"] == "hbm" and out["roofline_hbm"]["batch_per_gpu"] == 16384
    # the driver keeps an 8 KB tail of stdout: the line must fit with room to spare (round 3's 20.7 KB line came back parsed = null),
    # everything else is in the side file the line names
    assert len(lines[0]) <= 4096, len(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and "workload" in out["config"]
    det = json.load(open(detail))
    assert det["value"] == out["value"] and "accounting" in det and all(v["n_gpus"] == 2 for v in det["other_configs"].values())
    # both ranks ran the same step sequence (warm-up + graph-build round + probe + R rounds for the headline)
    per_rank = [open("%s.%d" % (log, k)).read().split() for k in range(2)]
    assert per_rank[0] == per_rank[1] and int(per_rank[0][0]) == 5 + 20 + 20 + 20 * R


def test_single_rank_default_path(tmp_path):
    r = run_bench(["--steps", "10", "--warmup", "2", "--min-region-ms", "10", "--no-extras"], {"DS_BENCH_DETAIL": str(tmp_path / "d.json")})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and "other_configs" not in out and "cpu_baseline" not in out and out["collective"] == "none"


def test_forced_process_group_at_world_size_one(tmp_path):
    # DS_DIST_FORCE=1: the collective path with ONE rank (what tests/test_gpu_bench.py runs over RCCL on a one-GPU box), here over gloo
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = run_bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--min-region-ms", "10", "--no-extras"],
                  {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "DS_DIST_FORCE": "1",
                   "DS_DIST_BACKEND": "gloo", "DS_BENCH_DETAIL": str(tmp_path / "d.json")})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["collective"] == "gloo"


def test_eight_ranks_shard_cfg4(tmp_path):
    # the launch form of the 8-GPU node: 8 ranks, BASELINE config 4's 8192 utterances sharded over them, the cross-rank verification on
    r = run_bench(["--gpus", "8", "--config", "cfg4", "--total-batch", "8192", "--steps", "4", "--warmup", "1", "--min-region-ms", "10", "--no-extras"],
                  {"DS_BENCH_DETAIL": str(tmp_path / "d.json")})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["batch_per_gpu"] == 1024 and out["verify"] == "ok"
    assert abs(out["value"] - 8192 * out["timed_steps"] / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3


def test_rank_count_mismatch_is_an_error():
    # launched as ONE rank (as torchrun would with nproc-per-node 1) but asked for 8 GPUs: must not bench one GPU and call it eight
    r = run_bench(["--gpus", "8", "--steps", "5", "--warmup", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "--gpus 8" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_torchrun_style_launch(tmp_path):
    # the other launch form of the contract: ranks started by torch.distributed.run
    env = dict(os.environ, DS_BENCH_BACKEND="tests.bench_stub:StubBackend", PYTHONPATH=ROOT, DS_BENCH_DETAIL=os.devnull)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                        "--min-region-ms", "10", "--no-extras"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2


def test_profiled_process_never_starts_child_ranks():
    # under rocprofv3 the preloaded tool has initialised the GPU before main(): bench.py must refuse to self-launch ranks
    r = run_bench(["--gpus", "2", "--steps", "5", "--warmup", "1"], {"ROCPROF_OUTPUT_PATH": "/tmp/x", "ROCPROFILER_LIBRARY_CTOR": "1"})
    assert r.returncode == 2 and "profiler" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "profile_bench.sh"), "t", "--gpus", "8"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "refused" in r.stderr
