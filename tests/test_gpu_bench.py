"""bench.py's rank path on real hardware without a multi-GPU node: `--gpus 2` with both ranks on GPU 0 (DS_FORCE_DEVICE=0, gloo) runs the
real GpuBackend through self-launch, barrier, MAX-over-ranks bracketing, the one reduce and the one JSON line; `--total-batch` shards one
job over the ranks with dist.shard_range (strong scaling).  Not a scaling measurement: the line says so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, tmp_path, extra_env=None, drop=()):
    env = dict(os.environ, DS_FORCE_DEVICE="0", DS_DIST_BACKEND="gloo", DS_BENCH_DETAIL=str(tmp_path / "detail.json"), PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DS_BENCH_BACKEND", "DS_DIST_FORCE") + tuple(drop):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-2000:]
    return json.loads(lines[0]), json.load(open(tmp_path / "detail.json"))


def test_two_ranks_share_one_gpu_weak_scaling(tmp_path):
    out, det = run_bench(["--gpus", "2", "--steps", "5", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--min-region-ms", "20"], tmp_path)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["data"] == "synthetic" and out["collective"] == "gloo"
    assert "share GPU 0" in out["config"]["note"]
    assert out["config"]["batch_per_gpu"] == 1024
    frames = 2 * 1024 * out["timed_steps"]                            # both ranks' frames over the slower rank's region
    assert abs(out["value"] - frames / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3
    assert out["roofline"]["bound"] == "hbm" and 0.0 < out["roofline"]["frac"] < 1.0
    assert det["value"] == out["value"]
    # every rank ran utterances [0, 8) as well; their checksums of samples and exported state agree (bench.py cross_rank_verify)
    assert out["verify"] == "ok" and out["config"]["verify"].startswith("2 rank(s)") and "affinity" in out["config"]


def test_two_ranks_shard_one_job_strong_scaling(tmp_path):
    """cfg4 worded as BASELINE words it — one job of N utterances sharded over the ranks: rank r holds utterances shard_range(N, r, 2)"""
    out, det = run_bench(["--gpus", "2", "--config", "cfg4", "--total-batch", "96", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                          "--min-region-ms", "20"], tmp_path)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["total_batch"] == 96 and out["config"]["batch_per_gpu"] == 48
    frames = 96 * out["timed_steps"]
    assert abs(out["value"] - frames / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3


def test_eight_ranks_shard_cfg4_as_baseline_words_it(tmp_path):
    """The launch form of the node this was never run on: 8 ranks (all on GPU 0 here, gloo), BASELINE config 4's 8192 utterances sharded over
    them with dist.shard_range, the cross-rank checksum comparison on, one line from rank 0."""
    out, det = run_bench(["--gpus", "8", "--config", "cfg4", "--total-batch", "8192", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                          "--min-region-ms", "10"], tmp_path)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["total_batch"] == 8192 and out["config"]["batch_per_gpu"] == 1024
    assert out["verify"] == "ok" and out["config"]["verify"].startswith("8 rank(s)")
    frames = 8192 * out["timed_steps"]
    assert abs(out["value"] - frames / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3


def test_verify_flag_on_one_rank(tmp_path):
    out, det = run_bench(["--gpus", "1", "--verify", "--steps", "5", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--min-region-ms", "20"], tmp_path)
    assert out["verify"] == "ok" and out["config"]["verify"].startswith("1 rank(s)")


def test_rccl_branch_runs_at_world_size_one(tmp_path):
    """dist.py's RCCL branch (backend "nccl": set_device before init_process_group, device-tensor all_reduce, barrier(device_ids)) on the one
    GPU there is: bench.py as the single rank of a torchrun-style environment with the process group forced into existence (DS_DIST_FORCE=1).
    The line names the collective it ran over."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               DS_DIST_FORCE="1", DS_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out, det = run_bench(["--gpus", "1", "--steps", "5", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--min-region-ms", "20"], tmp_path,
                         extra_env=env, drop=("DS_FORCE_DEVICE",))
    assert out["collective"] == "nccl" and out["n_gpus"] == 1
    frames = 1024 * out["timed_steps"]
    assert abs(out["value"] - frames / (out["region_ms"] * 1e-3)) / out["value"] < 1e-3
    assert out["roofline"]["bound"] == "hbm" and 0.0 < out["roofline"]["frac"] < 1.0
