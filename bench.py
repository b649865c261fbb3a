#!/usr/bin/env python3
"""bench.py — enhanced frames/s of the MVDR hot path (BASELINE.json configs[1]):
adaptive MVDR, 4 mics, 16 kHz, 512-FFT / 256-hop, batch = 1024 synthetic utterances per GPU.

A "step" is one pass of the hot path over the batch in the streaming-callback regime: ONE hop
(256 samples x 4 channels) of every utterance in -> one hop of enhanced audio out, all carried
state (STFT tail, OLA tail, Rvv, MCRA trackers) read from and written back to HBM (T = 1 in
SURVEY.md section 8d).  Inputs are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps 625 --warmup 25
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

M, NFFT, HOP, BATCH, FS = 4, 512, 256, 1024, 16000
ANGLE = np.array([197.0, 0.0]) / 180.0 * np.pi
# algorithmic bytes per frame at T hops per call
#   bytes(T) = M*hop*4 (in) + hop*4 (out) + 2*S/T,  S = 43 156 B of output-affecting state per utterance
S_STATE = 4096 + 1024 + 257 * 16 * 8 + 5 * 257 * 4


# cfg3 (GSC + McMcra gain): S = 5120 tails + G_aic 3*257*8 + Phi_yy, Phi_vv 2*16*257*4 = 44 184 B ; cfg1 (fixed): tails only
S_STATE_BY_ALGO = {"mvdr": S_STATE, "gsc": 5120 + 3 * 257 * 8 + 2 * 16 * 257 * 4, "fixed": 5120}


def algorithmic_bytes_per_frame(T, algo="mvdr"):
    return M * HOP * 4 + HOP * 4 + 2.0 * S_STATE_BY_ALGO[algo] / T


HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_batch_torch(torch, B, L, device, seed):
    """[B, M, L] float32 on `device`: white noise sigma=0.05 per mic + 0.5 s on/off band-limited
    (300-3400 Hz) Gaussian source sigma=0.1 steered from 197 deg through the array delays
    (BASELINE.md section 3), generated on the GPU."""
    from distantspeech_amd.mic_array import MicArray, compute_tau
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    mic = MicArray(M=M, n_fft=NFFT)
    tau = torch.tensor(compute_tau(mic, ANGLE)[:, 0], dtype=torch.float32, device=device)
    x = torch.empty((B, M, L), dtype=torch.float32, device=device)
    f = torch.fft.rfftfreq(L, 1.0 / FS).to(device)
    band = ((f >= 300) & (f <= 3400)).to(torch.float32)
    gate = ((torch.arange(L, device=device) // (FS // 2)) % 2 == 0).to(torch.float32)
    chunk = 64
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        S = torch.fft.rfft(torch.randn((b1 - b0, L), generator=g, device=device)) * band
        for m in range(M):
            ph = torch.exp(-2j * math.pi * f * tau[m])
            s = torch.fft.irfft(S * ph, n=L)
            s = s / (s.std(dim=1, keepdim=True) + 1e-12) * 0.1
            x[b0:b1, m] = s * gate + 0.05 * torch.randn((b1 - b0, L), generator=g, device=device)
    return x


# ------------------------------------------------------------------------------------------------
# CPU baseline leg: the oracle (a "port": plain-C double-precision restatement of the reference's loop,
# oracle/c/ds_oracle_mvdr.c, pinned to the reference's golden vectors) on the host cores, one thread per core
# ------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s=10.0):
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ds_oracle as O
    from oracle.c_oracle import COracleMVDR
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))                   # threads actually used (reported as `cores`)
    mic = O.OracleMicArray(M=M, n_fft=NFFT)
    tao = O.circular_tao(mic.r, mic.c, mic.gamma, ANGLE)
    omega = 2 * np.pi * np.arange(NFFT // 2 + 1) * FS / NFFT
    steer = np.exp(-1j * omega[:, None] * tao[None, :])
    n_distinct = 8

    def work(args):                                  # ctypes releases the GIL: threads run on separate cores
        x, = args
        eng = COracleMVDR(steer, NFFT, HOP)
        t0 = time.perf_counter()
        eng.process(x)                               # hop-by-hop inside (one hop per reference call)
        return x.shape[1] // HOP, time.perf_counter() - t0

    f0, t0 = work((O.synth_utterance(0, HOP * 200, mic),))      # calibrate on one core, then ~budget_s/4 of work per thread
    frames = int(max(200, min(625 * 4, f0 / t0 * budget_s / 4)))
    xs = [O.synth_utterance(u, HOP * frames, mic) for u in range(n_distinct)]   # inputs built before the timed region
    t_start = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        res = list(pool.map(work, [(xs[u % n_distinct],) for u in range(cores)]))
    wall = time.perf_counter() - t_start
    busy = max(r[1] for r in res)
    total = sum(r[0] for r in res)
    return {"value": round(total / busy, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d streams x %d hops (one hop per call; %d distinct synthetic utterances), oracle/c/ds_oracle_mvdr.c "
                      "(plain C, fp64), one thread per core; %.1f s wall" % (cores, frames, n_distinct, wall),
            "per_core": round(total / busy / cores, 1)}


def load_traffic():
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=625)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU (BASELINE cfg2: 1024)")
    ap.add_argument("--hops-per-step", type=int, default=1, help="T: hops per call (1 = streaming callback regime)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algo", default="mvdr", choices=["mvdr", "gsc", "fixed"],
                    help="mvdr = BASELINE cfg2 (the headline); gsc = cfg3 (use --batch 4096); fixed = cfg1 on the GPU")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE config: cfg2 = the headline (default); cfg3 = --algo gsc --batch 4096; cfg4 / cfg5 = the chain handles "
                         "(scripts/bench_cfg4.py, scripts/bench_cfg5.py: single GPU, one JSON line per regime)")
    args = ap.parse_args()
    if args.config in ("cfg4", "cfg5"):
        import runpy
        sys.argv = [sys.argv[0]]
        runpy.run_path(os.path.join(ROOT, "scripts", "bench_%s.py" % args.config), run_name="__main__")
        return
    if args.config == "cfg3":
        args.algo = "gsc"
        if args.batch == BATCH:
            args.batch = 4096

    import torch
    from distantspeech_amd import BatchEngine, dist as dsdist
    from distantspeech_amd import _lib as L
    from distantspeech_amd.mic_array import MicArray

    rank, local_rank, world = dsdist.init()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    B, K, W, T = args.batch, args.steps, args.warmup, args.hops_per_step
    hops_total = (K + W) * T
    Ltot = hops_total * HOP
    x = synth_batch_torch(torch, B, Ltot, device, seed=rank)
    y = torch.empty((B, Ltot), dtype=torch.float32, device=device)

    algo_id = {"mvdr": L.ALGO_ADAPTIVE, "gsc": L.ALGO_GSC, "fixed": L.ALGO_FIXED}[args.algo]
    eng = BatchEngine(algo_id, M, NFFT, HOP, batch=B, device=local_rank)
    mic = MicArray(M=M, n_fft=NFFT)
    tao = -1 * mic.r * np.cos(ANGLE[1]) * np.cos(ANGLE[0] - mic.gamma) / mic.c
    omega = 2 * np.pi * np.arange(NFFT // 2 + 1) * FS / NFFT
    a = np.exp(-1j * omega[:, None] * tao[None, :])
    eng.set_steering(a / M if args.algo == "fixed" else a)          # fixed: delay-and-sum weights W = a / M
    if args.algo != "fixed":
        eng.set_method(L.METHOD_MVDR)

    xp, yp = x.data_ptr(), y.data_ptr()
    torch.cuda.synchronize()          # inputs resident before anything is launched on the engine's stream

    def run_steps(first_step, n):
        """n successive steps (one native call; launches go to the engine's own HIP stream)."""
        off = first_step * T * HOP
        eng.process_device_seq(xp + 4 * off, L.LAYOUT_CHANNELS_SAMPLES, M * Ltot, Ltot, T * HOP, T * HOP, n,
                               yp + 4 * off, Ltot, T * HOP)

    run_steps(0, W)
    eng.synchronize()
    torch.cuda.synchronize()
    dsdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.timing_begin()                # hipEvent on the stream the kernels are launched on
    run_steps(W, K)
    dev_ms = eng.timing_end()         # records the end event and waits for it
    torch.cuda.synchronize()
    dsdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    frames_rank = B * K * T
    frames, t_max = dsdist.reduce_throughput(frames_rank, elapsed, device=device)
    assert bool(torch.isfinite(y[:, W * T * HOP:]).all()), "non-finite output"

    if rank == 0:
        launch_ms = dev_ms / K                                  # average launch duration (HIP events, same stream)
        alg_bytes = algorithmic_bytes_per_frame(T, args.algo) * B * T      # per launch
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        out = {
            "metric": "enhanced frames/sec (4-mic, 512-FFT)", "value": round(frames / t_max, 1), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(t_max / K * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, 4 mics, 16 kHz, 512-FFT/256-hop, "
                                   "batch=%d utterances per GPU, %d hop(s) per call (streaming callback regime), "
                                   "state resident in HBM" % (
                                       {"mvdr": "adaptive MVDR (adaptivebeamfomer.process method=2)",
                                        "gsc": "GSC + LMS canceller + McMcra gain (GSC.process method=2)",
                                        "fixed": "delay-and-sum (FixedBeamformer.process)"}[args.algo], B, T),
                       "batch_per_gpu": B, "hops_per_call": T, "n_mics": M, "nfft": NFFT, "hop": HOP},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic() if (args.algo == "mvdr" and B == BATCH and T == 1) else None,
                         "kernel": "ds_frames_kernel<512,4,%s>" % {"mvdr": "ADAPTIVE", "gsc": "GSC", "fixed": "FIXED"}[args.algo], "launch_ms": round(launch_ms, 5),
                         "algorithmic_bytes_per_frame": algorithmic_bytes_per_frame(T, args.algo)},
        }
        if world == 1 and not args.no_cpu_baseline and args.algo == "mvdr":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))


if __name__ == "__main__":
    main()
