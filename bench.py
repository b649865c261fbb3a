#!/usr/bin/env python3
"""bench.py — enhanced frames/s of the per-frame enhancement hot path on MI355X.

Headline = BASELINE.json configs[1]: adaptive MVDR, 4 mics, 16 kHz, 512-FFT / 256-hop, batch = 1024 synthetic utterances
per GPU.  A "step" is one pass of the hot path over the batch in the streaming-callback regime: ONE hop (256 samples x M
channels) of every utterance in -> one hop of enhanced audio out, all carried state read from and written back to HBM (T = 1
in SURVEY.md section 8d).  Inputs are resident in HBM before the timed region starts.

    python bench.py                                   # 1 GPU, defaults finish within minutes
    python bench.py --gpus 8 --steps K --warmup W     # starts 8 child ranks itself (one per GPU, RCCL), rank 0 prints the line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # or launched as ranks by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the env)

Timed region: W untimed warm-up steps, then R rounds of EXACTLY K steps each, enqueued back to back and bracketed ONCE by
barrier + device synchronize on both sides; R is the smallest count that makes the region >= --min-region-ms (the same on
every rank), so a short --steps still gives a region the driver's clock can see.  `value` = frames of all ranks / MAX-over-ranks
wall time of the region; `ms_per_step` = that time / (R * K).  HIP events on the engine's stream give the per-launch duration
the roofline uses.

ONE JSON line of at most 4 KB on rank 0 (contract in the task statement): the contract keys, a numeric `roofline` (+ `roofline_hbm`: the
same kernel with the state working set outside the 256 MiB Infinity Cache), `cpu_baseline`, and per other BASELINE config (cfg3 / cfg4 /
cfg5 at one hop per call, every config with 10 s of signal per call, the wide-tap WPE) `{value, ms_per_step, bound, frac}`.  Everything
else — accounting prose, per-config CPU legs, the latency block, sources of the profile-derived figures — goes to the side file the line
names in `detail` (default bench_detail.json next to this script; DS_BENCH_DETAIL=<path> overrides).  A --gpus / rank-count mismatch exits
non-zero."""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 16000
LEAD_HOPS = 16            # one-hop-per-call workloads: hops of the utterance in front of the resident slab (GpuWorkload)
ANGLE_DEG = (197.0, 0.0)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic bytes per frame at T hops per call
#   bytes(T) = M*hop*4 (in) + hop*4 (out) + 2*S/T,   S = output-affecting persistent state per utterance (fp32 / complex64)
WORKLOADS = {
    # cfg2 (headline): S = STFT tail 4096 + OLA tail 1024 + Rvv 257*16*8 + MCRA 5*257*4 = 43 156
    "cfg2": dict(algo="ADAPTIVE", M=4, nfft=512, hop=256, batch=1024, S=4096 + 1024 + 257 * 16 * 8 + 5 * 257 * 4, r=0.032,
                 kernel="ds_frames_kernel<512,4,ADAPTIVE>", launches=1, graph=1,
                 desc="adaptive MVDR (adaptivebeamfomer.process method=2), 4 mics, 16 kHz, 512-FFT/256-hop"),
    # cfg3: S = 5120 tails + G_aic 3*257*8 + Phi_yy, Phi_vv 2*16*257*4 = 44 184
    "cfg3": dict(algo="GSC", M=4, nfft=512, hop=256, batch=4096, S=5120 + 3 * 257 * 8 + 2 * 16 * 257 * 4, r=0.032,
                 kernel="ds_frames_kernel<512,4,GSC> (two utterance groups of 2048 on two streams)", launches=2, graph=1,
                 desc="GSC + LMS canceller + McMcra gain (GSC.process method=2), 4 mics, 16 kHz, 512-FFT/256-hop"),
    # the north star's target sentence ("4-mic MVDR + postfilter ... frames/s at >= 40 % HBM roofline"): cfg2's beamformer with the McMcra gain of the
    # same frame on its output (GSC.py:225,286), ONE fused frame kernel (DS_ALGO_ADAPTIVE_PF).  S = cfg2's 43 156 + Phi_yy, Phi_vv 2*16*257*4
    "mvdr_pf": dict(algo="ADAPTIVE_PF", M=4, nfft=512, hop=256, batch=1024, S=4096 + 1024 + 257 * 16 * 8 + 5 * 257 * 4 + 2 * 16 * 257 * 4, r=0.032,
                    kernel="ds_frames_kernel<512,4,ADAPTIVE_PF>", launches=1, graph=1,
                    desc="adaptive MVDR + McMcra post-filter gain in one pass (adaptivebeamfomer.process method=2, * spp.G), 4 mics, 16 kHz, 512-FFT/256-hop"),
    # the only maintained MVDR usage of the reference (example/mvdr.ipynb cell 4, the flow its published PESQ 2.26 was obtained with, 6 microphones):
    # Transform.stft -> McSpp.estimation -> steering(Phi_xx) -> compute_mvdr_weight(steer, Phi_vv_inv) -> w^H y -> Transform.istft as ONE handle
    # (DS_ALGO_MCSPP_MVDR).  S = tails (M + 1) * 256 * 4 + 257 * (Phi_yy, Phi_vv complex M x M: 2 * M * M * 8 + McCDR / MCRA / xi, gamma, p rows 12 * 4)
    # (batch 4096 at one hop per call since the end of round 6: the operator's launch at 1024 utterances is 1056 workgroups for 1024 slots — 11.4 M /
    # 22.5 M frames/s at 1024, 13.9 M / 26.1 M at 2048, 15.0 M / 29.2 M at 4096; the 10 s regime runs 2048 — 17.7 M at 1024, 20.2 M at 2048, 21.7 M at 4096 with 32 GB of spectra)
    "nb_mvdr": dict(algo="MCSPP_MVDR", M=6, nfft=512, hop=256, batch=4096, batch_chunked=2048, S=7 * 256 * 4 + 257 * (2 * 36 * 8 + 48), r=0.05,
                    kernel="DS_ALGO_MCSPP_MVDR chain: ds_binop_kernel<MCSPP,6> (+ analysis, McCDR, synthesis)", launches=4, graph=1,
                    desc="online MVDR of mvdr.ipynb cell 4 (McSpp + steering + MVDR weights per frame), 6 mics, 16 kHz, 512-FFT/256-hop"),
    "nb_mvdr_m4": dict(algo="MCSPP_MVDR", M=4, nfft=512, hop=256, batch=4096, batch_chunked=2048, S=5 * 256 * 4 + 257 * (2 * 16 * 8 + 48), r=0.032,
                       kernel="DS_ALGO_MCSPP_MVDR chain: ds_binop_kernel<MCSPP,4> (+ analysis, McCDR, synthesis)", launches=4, graph=1,
                       desc="online MVDR of mvdr.ipynb cell 4 (McSpp + steering + MVDR weights per frame), 4 mics, 16 kHz, 512-FFT/256-hop"),
    # cfg1 on the GPU (stateless apart from the tails)
    "fixed": dict(algo="FIXED", M=4, nfft=512, hop=256, batch=1024, S=5120, r=0.032,
                  kernel="ds_frames_kernel<512,4,FIXED>", launches=1, graph=1,
                  desc="delay-and-sum (FixedBeamformer.process), 4 mics, 16 kHz, 512-FFT/256-hop"),
    # cfg4: WPE (N = 2 taps, delay 4) -> adaptive MVDR -> SPP gain; 8 mics, 1024/512; 8192 utterances over 8 GPUs = 1024 per GPU
    "cfg4": dict(algo="WPE_MVDR", M=8, nfft=1024, hop=512, batch=1024, S=2197656, r=0.05, filter_len=2,
                 kernel="DS_ALGO_WPE_MVDR chain", launches=5, graph=0,     # plain launches: a captured step joins the two utterance groups at its end (+3 %, profiles/r06a/graph_ab.txt)
                 desc="WPE dereverberation (2 taps) + adaptive MVDR + SPP gain chain, 8 mics, 16 kHz, 1024-FFT/512-hop"),
    # cfg5: SubbandGSC structure with SubbandRLS blocking filters; 6 mics, 512 bands; 16384 utterances over 8 GPUs = 2048 per GPU
    "cfg5": dict(algo="SUBBAND_GSC", M=6, nfft=512, hop=256, batch=2048, S=313440, r=0.05, filter_len=2, rls_lambda=0.998,
                 kernel="DS_ALGO_SUBBAND_GSC chain", launches=7, graph=0,     # plain launches: hipGraph replay serialises the chain's streams
                 desc="Subband-RLS GSC chain (SubbandGSC.process with SubbandRLS blocking filters), 6 mics, 16 kHz, 512 bands / block 256"),
    # the reference's maintained use of Wpe (example/wpe.ipynb cell 2): Wpe(channels=4, filter_len=20, delay=4, num_bands=256, hop_length=64),
    # ONE call per hop (DS_ALGO_WPE_TD: analysis -> delay line -> RLS-WPE, one wavefront per (utterance, bin) -> synthesis of channel 0).
    # S = K (CN (CN + 1) / 2 + C CN + CN) 8 B (P as its Hermitian triangle, W, taps) = 129 * 3640 * 8 + the delay line 4 * 129 * 4 * 8 + tails
    "wpe_nb": dict(algo="WPE_TD", M=4, nfft=256, hop=64, batch=1024, S=129 * 3640 * 8 + 4 * 129 * 4 * 8 + 4 * 192 * 4 + 192 * 4, r=0.032, filter_len=20,
                   rls_lambda=0.998, kernel="ds_wpe_wide_kernel<80,2,4,20> (+ analysis, synthesis)", launches=3, graph=1,
                   desc="Wpe.update at the notebook's operating point (4 channels x 20 taps, delay 4, 256 bands / hop 64), dereverberated channel 0"),
    # SURVEY 8(d)'s 10-tap sizing of BASELINE config 4: the cfg4 chain with 8 x 10 = 80 taps-by-channels (S per the survey's formula with the
    # covariance unpacked: 513 (80^2 + 8 * 80 + 80) 8 B = 29.2 MB + the cfg4 rest); 256 utterances per GPU = 4.3 GB of state
    "cfg4_n10": dict(algo="WPE_MVDR", M=8, nfft=1024, hop=512, batch=256, S=513 * (6400 + 640 + 80) * 8 + 18432 + 262656 * 2 + 10260, r=0.05, filter_len=10,
                     kernel="DS_ALGO_WPE_MVDR chain, ds_wpe_wide_kernel<80,2,8,10>", launches=5, graph=1,
                     desc="WPE dereverberation (10 taps) + adaptive MVDR + SPP gain chain, 8 mics, 16 kHz, 1024-FFT/512-hop"),
    # the two overlap-save GSCs as chain handles, 4 mics, block 256; not BASELINE configs (--config tdgsc / fdgsc).
    # S (fp32): TDGSC = canceller W 3*257*8 + P 257*4 + previous input block 3*256*4 + non-causal delay 128*4, MCRA 5*257*4, analysis tail
    # 1024, FIR history 83*4*4, notch 32 = 18 304; FDGSC = 4 blocking filters (257*8 + 257*4 + 1024) + canceller (4*257*8 + 257*4 + 4096),
    # MCRA 5140, analysis tails 2*1024, FIR history 1328, notch 32, delays 4*128*4 + 1024 = 39 316
    # (batch 4096 since the end of round 6: every kernel of these chains is one short dependent program per utterance, so at 1024 utterances a
    # launch is a single round of small workgroups — 15.5 M / 9.7 M frames/s against 24.5 M / 13.6 M at 4096, 26.0 M / 14.3 M at 8192)
    "tdgsc": dict(algo="TDGSC", M=4, nfft=512, hop=256, batch=4096, S=18304, r=0.032, kernel="DS_ALGO_TDGSC chain", launches=6, graph=0,
                  desc="TDGSC chain (TDGSC.process: FIR bank + blocking matrix + MCRA-controlled overlap-save canceller), 4 mics, 16 kHz, block 256"),
    "fdgsc": dict(algo="FDGSC", M=4, nfft=512, hop=256, batch=4096, S=39316, r=0.032, kernel="DS_ALGO_FDGSC chain", launches=7, graph=0,
                  desc="FDGSC chain (FDGSC.process: adaptive blocking filters + norm-limited canceller), 4 mics, 16 kHz, block 256"),
}


def batch_of(w, hops_per_call):
    """utterances per GPU of a workload: `batch`, or `batch_chunked` for calls of more than one hop where the workload names one"""
    return w.get("batch_chunked", w["batch"]) if hops_per_call > 1 else w["batch"]


EXTRA_T1 = ("mvdr_pf", "cfg3", "cfg4", "cfg5", "wpe_nb", "nb_mvdr", "nb_mvdr_m4", "tdgsc", "fdgsc")                                                  # other_configs at one hop per call
EXTRA_CHUNKED = (("cfg2", 625), ("mvdr_pf", 625), ("cfg3", 625), ("cfg4", 312), ("cfg5", 625), ("wpe_nb", 2500), ("nb_mvdr", 625))         # ... and with 10 s per call
DATA_NOTE = ("BASELINE.md section 3's recipe in both legs (white noise sigma 0.05 per microphone + a 0.5 s on / off 300-3400 Hz Gaussian source sigma "
             "0.1 steered from 197 degrees, seed 1234 + utterance): the GPU leg draws it on the device with torch's generator (GpuBackend.synth), the "
             "cpu_baseline legs with NumPy's (oracle.synth_utterance) — the same statistics, different random streams; neither leg's arithmetic per "
             "frame depends on the sample values")


def algorithmic_bytes_per_frame(w, T=1):
    return w["M"] * w["hop"] * 4 + w["hop"] * 4 + 2.0 * w["S"] / T


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no rank environment starts N child ranks, one per GPU.  The parent never imports
# torch and never touches the GPU; the children are this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_profiler():
    """True when a GPU profiler's tool library is preloaded into this process (rocprofv3 / rocprof): the tool initialises the GPU before
    main() runs, so this process must not start another program — no child ranks, no CPU-baseline worker processes."""
    env = os.environ
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_TOOL", "ROCTRACER_")) for k in env):
        return True
    return any(t in env.get("LD_PRELOAD", "") for t in ("rocprofiler", "roctracer", "rocprof"))


def launch_ranks(n, argv):
    env0 = dict(os.environ)
    env0.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), DS_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:                     # one rank failed: stop exactly the ranks this process started
                    q.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------
# GPU backend: device buffers through torch (plumbing), the hot path through libdsenh.so's C-ABI (BatchEngine)
# ------------------------------------------------------------------------------------------------
class GpuBackend:
    name = "synthetic"
    dist_backend = None              # dist.init picks nccl (= RCCL) on a GPU box

    def __init__(self, local_rank, world, tail_async=False):
        import torch
        self.torch = torch
        self.tail_async = bool(tail_async)        # main() decides: it is the one that raised the runtime's hardware-queue limit (or did not)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
        # DS_FORCE_DEVICE=<ordinal> (with DS_DIST_BACKEND=gloo): every rank on that one GPU — exercises the real rank path (sharding, barrier,
        # reduce, one JSON line) on a single-GPU box; the line then says so in `config.note` and is not a scaling measurement
        forced = os.environ.get("DS_FORCE_DEVICE")
        dev = int(forced) if forced is not None else local_rank
        if torch.cuda.device_count() <= dev:
            raise SystemExit("rank with LOCAL_RANK=%d needs GPU %d but only %d GPU(s) are visible" % (local_rank, dev, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        self.local_rank = dev
        self.shared_device = forced is not None and world > 1
        self.device = torch.device("cuda", dev)
        if self.shared_device:
            self.dist_backend = "gloo"

    def device_sync(self):
        self.torch.cuda.synchronize()

    def synth(self, w, B, L, seed):
        """[B, M, L] float32 on the GPU: white noise sigma = 0.05 per mic + 0.5 s on/off band-limited (300-3400 Hz) Gaussian
        source sigma = 0.1 steered from 197 deg through the array delays."""
        import numpy as np
        torch = self.torch
        from distantspeech_amd.mic_array import MicArray, compute_tau
        M = w["M"]
        g = torch.Generator(device=self.device)
        g.manual_seed(1234 + seed)
        mic = MicArray(arrayType="circular", r=w["r"], M=M, n_fft=w["nfft"])
        ang = np.array(ANGLE_DEG) / 180.0 * np.pi
        tau = torch.tensor(compute_tau(mic, ang)[:, 0], dtype=torch.float32, device=self.device)
        x = torch.empty((B, M, L), dtype=torch.float32, device=self.device)
        if os.environ.get("DS_BENCH_SYNTH") == "white":      # profiling runs: the traffic does not depend on the content, skip the FFTs
            for b0 in range(0, B, 256):
                x[b0:b0 + 256] = 0.05 * torch.randn(x[b0:b0 + 256].shape, generator=g, device=self.device)
            return x
        f = torch.fft.rfftfreq(L, 1.0 / FS).to(self.device)
        band = ((f >= 300) & (f <= 3400)).to(torch.float32)
        gate = ((torch.arange(L, device=self.device) // (FS // 2)) % 2 == 0).to(torch.float32)
        chunk = 64
        for b0 in range(0, B, chunk):
            b1 = min(B, b0 + chunk)
            S = torch.fft.rfft(torch.randn((b1 - b0, L), generator=g, device=self.device)) * band
            for m in range(M):
                ph = torch.exp(-2j * math.pi * f * tau[m])
                s = torch.fft.irfft(S * ph, n=L)
                s = s / (s.std(dim=1, keepdim=True) + 1e-12) * 0.1
                x[b0:b1, m] = s * gate + 0.05 * torch.randn((b1 - b0, L), generator=g, device=self.device)
        return x

    def make(self, w, B, T, K, W, seed, graph, split=None):
        return GpuWorkload(self, w, B, T, K, W, seed, graph, split)

    def latency(self, n_chunks=1500, warm=100):
        """The reference's only service-level figure: the realtime shell must finish one 1024-sample chunk within its 64 ms
        (realtime/realtime_processing.py:113-136, CHUNK = 1024 at 16 kHz).  ONE stream (B = 1), 4 microphones, 512 / 256, adaptive MVDR:
        the shell's wire format through ds_process_pcm16 (int16 interleaved 6-channel frames in host memory, int16 out: PCIe both ways and
        two synchronisations inside the timed call) and the same chunk through ds_process_device (device-resident float chunk; timed with
        the stream synchronisation).  Median and p99 of the wall time per chunk."""
        import numpy as np
        from distantspeech_amd import BatchEngine, _lib as L
        from distantspeech_amd.mic_array import MicArray
        torch = self.torch
        w = WORKLOADS["cfg2"]
        M, nfft, hop, CH = w["M"], w["nfft"], w["hop"], 1024
        mic = MicArray(arrayType="circular", r=w["r"], M=M, n_fft=nfft)
        ang = np.array(ANGLE_DEG) / 180.0 * np.pi
        tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c
        a = np.exp(-1j * (2 * np.pi * np.arange(nfft // 2 + 1) * FS / nfft)[:, None] * tao[None, :])
        rng = np.random.default_rng(7)
        n = n_chunks + warm
        pcm = (rng.standard_normal((n, CH, 6)) * 1500).astype("<i2")                  # [chunk][sample][6 channels], microphones in 1..4
        out = {}
        eng = BatchEngine(L.ALGO_ADAPTIVE, M, nfft, hop, batch=1, device=self.local_rank)
        eng.set_steering(a); eng.set_method(L.METHOD_MVDR)
        t = np.empty(n)
        for i in range(n):
            t0 = time.perf_counter()
            eng.process_pcm16(pcm[i][None], first_channel=1)
            t[i] = time.perf_counter() - t0
        out["pcm16_host"] = t[warm:] * 1e6
        eng.close()
        eng = BatchEngine(L.ALGO_ADAPTIVE, M, nfft, hop, batch=1, device=self.local_rank)
        eng.set_steering(a); eng.set_method(L.METHOD_MVDR)
        xd = torch.tensor((pcm[:, :, 1:5].astype(np.float32) / 32768.0).transpose(0, 2, 1).copy(), device=self.device)   # [chunk][M][CH]
        yd = torch.empty((n, CH), dtype=torch.float32, device=self.device)
        self.device_sync()
        for i in range(n):
            t0 = time.perf_counter()
            eng.process_device(xd[i].data_ptr(), L.LAYOUT_CHANNELS_SAMPLES, M * CH, CH, yd[i].data_ptr(), CH)
            eng.synchronize()
            t[i] = time.perf_counter() - t0
        out["device"] = t[warm:] * 1e6
        eng.close()
        res = {"workload": "ONE stream (batch 1), adaptive MVDR, 4 mics, 512-FFT/256-hop, one 1024-sample chunk (4 hops) per call — the realtime "
                           "shell's callback (realtime/realtime_processing.py:113-136)", "chunk_samples": CH, "chunk_ms_budget": CH / FS * 1e3,
               "chunks_timed": n_chunks, "unit": "us per chunk (host wall clock around the call)"}
        for k, v in out.items():
            res[k] = {"median_us": round(float(np.median(v)), 1), "p99_us": round(float(np.percentile(v, 99)), 1), "max_us": round(float(v.max()), 1)}
        res["pcm16_host"]["path"] = "ds_process_pcm16: int16 interleaved 6-channel frames in host memory in, int16 out (H2D, conversion kernel, frame kernel, conversion kernel, D2H)"
        res["device"]["path"] = "ds_process_device + ds_synchronize: float chunk resident in HBM, output left in HBM"
        res["realtime_factor_p99"] = round(res["chunk_ms_budget"] * 1e3 / res["pcm16_host"]["p99_us"], 1)
        return res


def host_api_entry(be, long_calls=True):
    """The API the reference's users call, at batch: `adaptivebeamfomer.process(x_numpy)` (adaptivebeamformer.py:44-128) with x in HOST memory —
    every call uploads its chunk, runs the frame kernel and downloads the enhanced samples, so its rate is a PCIe figure.  B = 1024 streams,
    4 microphones, 512 / 256: one 1024-sample chunk per call (4 hops: the realtime shell's CHUNK, realtime/realtime_processing.py:113-136) and
    10 s per call (625 hops), through (a) the mirror class (float32 in, float64 out like the reference), (b) the C-ABI entry it sits on
    (ds_process: float32 both ways) and (c) the shell's wire format (ds_process_pcm16: int16 interleaved 6-channel frames in, int16 out).
    GB/s = (bytes up + bytes down) / wall time per call; pinned_h2d / d2h = this box's hipMemcpy rate from pinned host memory (16 MB)."""
    import numpy as np
    import distantspeech_amd as d
    from distantspeech_amd import _lib as L
    torch = be.torch
    B, M, nfft, hop = 1024, 4, 512, 256
    ang = np.array(ANGLE_DEG) / 180.0 * np.pi
    res = {"workload": host_api_entry.__doc__.split("\n")[0].strip(), "batch": B, "n_mics": M, "unit": "frames/s; GB/s = bytes both ways / wall time"}
    # the machine's own transfer rates (what `frac_of_pinned` is quoted against)
    pin = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(16 << 20, dtype=torch.uint8, device=be.device)
    for key, (dst, src) in (("pinned_h2d_gbs", (dev, pin)), ("pinned_d2h_gbs", (pin, dev))):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        res[key] = round(20 * (16 << 20) / (time.perf_counter() - t0) / 1e9, 2)
    del pin, dev
    rng = np.random.default_rng(3)
    for T in ((4, 625) if long_calls else (4,)):
        n = T * hop
        blk = (rng.standard_normal((M, 4096)) * 0.05).astype(np.float32)
        x = np.ascontiguousarray(np.broadcast_to(np.tile(blk, (1, -(-n // 4096)))[:, :n], (B, M, n)))      # [B, M, n] float32, 16 MB / 2.6 GB
        pcm = np.zeros((B, n, 6), dtype="<i2")
        pcm[:, :, 1:5] = (x[0].T * 32768).astype("<i2")[None]
        reps = 30 if T == 4 else 2
        ent = {}
        ab = d.adaptivebeamfomer(d.MicArray(arrayType="circular", r=0.032, M=M, n_fft=nfft), frameLen=nfft, batch=B, device=be.local_rank, track_ryy=False)
        for _ in range(2):
            ab.process(x, ang, method=2)
        t0 = time.perf_counter()
        for _ in range(reps):
            y = ab.process(x, ang, method=2)["data"]
        dt = (time.perf_counter() - t0) / reps
        ent["mirror"] = {"frames_s": round(B * T / dt, 1), "ms_per_call": round(dt * 1e3, 3), "gbs": round((x.nbytes + B * n * 4) / dt / 1e9, 2),
                         "path": "adaptivebeamfomer.process(x[B, M, n] float32 NumPy) -> dict, 'data' float64 like the reference's"}
        assert np.all(np.isfinite(y))
        del y
        ab.out_dtype = np.float32                       # the mirror's option: the kernel's own float32 samples, no conversion pass
        ab.process(x, ang, method=2)
        t0 = time.perf_counter()
        for _ in range(reps):
            y = ab.process(x, ang, method=2)["data"]
        dt = (time.perf_counter() - t0) / reps
        ent["mirror_f32"] = {"frames_s": round(B * T / dt, 1), "ms_per_call": round(dt * 1e3, 3), "gbs": round((x.nbytes + B * n * 4) / dt / 1e9, 2),
                             "frac_of_pinned": round((x.nbytes + B * n * 4) / dt / 1e9 / max(res["pinned_h2d_gbs"], 1e-9), 3),
                             "path": "the same with obj.out_dtype = np.float32"}
        eng = ab._eng
        for name, call, up, down in (("c_abi", lambda: eng.process(x, L.LAYOUT_CHANNELS_SAMPLES), x.nbytes, B * n * 4),
                                     ("pcm16", lambda: eng.process_pcm16(pcm, first_channel=1), pcm.nbytes, B * n * 2)):
            call()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            dt = (time.perf_counter() - t0) / reps
            ent[name] = {"frames_s": round(B * T / dt, 1), "ms_per_call": round(dt * 1e3, 3), "gbs": round((up + down) / dt / 1e9, 2),
                         "frac_of_pinned": round((up + down) / dt / 1e9 / max(res["pinned_h2d_gbs"], 1e-9), 3)}
        ent["c_abi"]["path"] = "ds_process: float32 [B, M, n] in host memory in, float32 [B, n] out"
        ent["pcm16"]["path"] = "ds_process_pcm16: int16 [B, n, 6] interleaved in host memory in (microphones 1..4), int16 [B, n] out"
        ent["mirror"]["frac_of_pinned"] = round(ent["mirror"]["gbs"] / max(res["pinned_h2d_gbs"], 1e-9), 3)
        ab._eng.close()
        res["T%d" % T] = ent
        del x, pcm, y
    return res


class GpuWorkload:
    """One engine handle + its resident input / output slabs; run(first_step, n) enqueues n successive steps."""

    def __init__(self, be, w, B, T, K, W, seed, graph, split=None):
        import numpy as np
        from distantspeech_amd import BatchEngine, _lib as L
        from distantspeech_amd.mic_array import MicArray, compute_tau
        torch = be.torch
        self.be, self.w, self.B, self.T, self.L = be, w, B, T, L
        M, nfft, hop = w["M"], w["nfft"], w["hop"]
        self.hop, self.M = hop, M
        # one hop per call: the resident slab starts LEAD_HOPS hops into the utterance, so that the K timed hops straddle the recipe's first
        # source-on -> source-off edge (0.5 s = hop 31.25 at hop 256) instead of sitting inside the first source-on segment, where the
        # VAD-gated noise-covariance update is mostly idle
        self.lead = LEAD_HOPS * hop if T == 1 else 0
        self.Ltot = (K + W) * T * hop
        xfull = be.synth(w, B, self.Ltot + self.lead, seed)
        self.x = xfull[:, :, self.lead:].contiguous() if self.lead else xfull
        del xfull
        self.y = torch.empty((B, self.Ltot), dtype=torch.float32, device=be.device)
        self.graph = graph
        algo = getattr(L, "ALGO_" + w["algo"])
        self.eng = BatchEngine(algo, M, nfft, hop, batch=B, device=be.local_rank, filter_len=w.get("filter_len", 0),
                               rls_lambda=w.get("rls_lambda", 0.0))
        if split is not None:
            self.eng.set_split(split)         # utterance groups of the fused frame kernels (default: 2 from 2048 utterances up)
        mic = MicArray(arrayType="circular", r=w["r"], M=M, n_fft=nfft)
        ang = np.array(ANGLE_DEG) / 180.0 * np.pi
        if w["algo"] == "WPE_TD":
            self.eng.set_wpe_delay(4)                                                       # awpe.py:36 / wpe.ipynb cell 2: delay=4
        elif w["algo"] == "MCSPP_MVDR":
            from distantspeech_amd.ops import McSpp
            self.eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(M, nfft))
        elif w["algo"] in ("SUBBAND_GSC", "TDGSC", "FDGSC"):
            from distantspeech_amd.ops import McSpp
            from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
            tau = compute_tau(mic, ang)
            self.eng.chain_set_aux(L.CHAIN_AUX_FIR, fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic.fs))
            if w["algo"] == "SUBBAND_GSC":
                self.eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(M, nfft))
                # the chain's tail on its own stream: only where this process raised the runtime's hardware-queue limit itself — main() says
                # so through the backend (tail_async); under a profiler the runtime was up before main() with its default 4 queues
                self.tail_async = bool(getattr(be, "tail_async", False))
                self.eng.set_param_i(L.PARAM_TAIL_ASYNC, int(self.tail_async))
        else:
            tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c          # adaptivebeamformer.py:52
            a = np.exp(-1j * (2 * np.pi * np.arange(nfft // 2 + 1) * FS / nfft)[:, None] * tao[None, :])
            self.eng.set_steering(a / M if w["algo"] == "FIXED" else a)                    # fixed: delay-and-sum weights W = a / M
            if w["algo"] != "FIXED":
                self.eng.set_method(L.METHOD_MVDR)
        be.device_sync()              # inputs resident before anything is launched on the engine's stream

    def state_bytes(self):
        """carried state of the handle as the library packs it (Hermitian / symmetric matrices as triangles), without the blob's header words"""
        return self.eng.state_bytes()

    def chain_min_bytes(self):
        """per-kernel byte budget of one step of a chain handle (cfg4 / cfg5), 0 for single-kernel workloads"""
        key = {"WPE_MVDR": "cfg4", "SUBBAND_GSC": "cfg5"}.get(self.w["algo"])
        if key is None or self.w is not WORKLOADS[key]:
            return 0
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        try:
            import stage_budget
        finally:
            sys.path.pop(0)
        return stage_budget.minimal_step_bytes(key, self.w, self.eng, self.B)

    def run(self, first_step, n):
        L, T, hop, Ltot = self.L, self.T, self.hop, self.Ltot
        off = 4 * first_step * T * hop
        self.eng.process_device_seq(self.x.data_ptr() + off, L.LAYOUT_CHANNELS_SAMPLES, self.M * Ltot, Ltot, T * hop, T * hop, n,
                                    self.y.data_ptr() + off, Ltot, T * hop, graph=self.graph)

    def sync(self):
        self.eng.synchronize()

    def timing_begin(self):
        self.eng.timing_begin()       # hipEvent on the stream the kernels are launched on

    def timing_end(self):
        return self.eng.timing_end()  # records the end event on that stream and waits for it

    def check(self, first_step):
        assert bool(self.be.torch.isfinite(self.y[:, first_step * self.T * self.hop:]).all()), "non-finite output"

    def checksums(self):
        """[64-bit sum of the output samples' bit patterns (position-weighted), the same of the exported state]: equal inputs through equal
        kernels give equal words, on any rank"""
        import numpy as np
        torch = self.be.torch
        bits = self.y.contiguous().view(torch.int32).to(torch.int64).flatten()
        wts = (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 65521) + 1
        ysum = int((bits * wts).sum().item())
        blob = np.frombuffer(self.eng.export_state(), dtype=np.uint8)
        pad = (-blob.size) % 4
        words = np.frombuffer(np.concatenate([blob, np.zeros(pad, np.uint8)]).tobytes(), dtype=np.uint32).astype(np.uint64)
        ssum = int((words * ((np.arange(words.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1))).sum(dtype=np.uint64).astype(np.int64))
        return [ysum, ssum]

    def close(self):
        self.eng.close()
        del self.x, self.y
        self.be.torch.cuda.empty_cache()


def load_backend(local_rank, world, tail_async=False):
    spec = os.environ.get("DS_BENCH_BACKEND")            # test infrastructure only (tests/bench_stub.py): exercises the rank plumbing without a GPU
    if spec:
        mod, cls = spec.split(":")
        return getattr(importlib.import_module(mod), cls)(local_rank, world)
    return GpuBackend(local_rank, world, tail_async=tail_async)


# ------------------------------------------------------------------------------------------------
# the measurement of one workload: warm-up, probe, R rounds of exactly K steps in ONE bracketed region
# ------------------------------------------------------------------------------------------------
def measure(be, dsdist, w, B, T, K, W, rank, world, min_region_ms, graph=None, max_rounds=20000, split=None, seed=None):
    wl = be.make(w, B, T, K, W, seed=rank if seed is None else seed, graph=w["graph"] if graph is None else graph, split=split)
    wl.run(0, W)
    wl.run(W, K)                      # one more untimed round: builds the round's hipGraph where one is used
    wl.sync()
    be.device_sync()
    t0 = time.perf_counter()          # probe round (untimed for the result): sizes R, the same on every rank
    wl.run(W, K)
    wl.sync()
    probe = time.perf_counter() - t0
    probe = dsdist.reduce_max(probe, device=getattr(be, "device", None))
    R = int(min(max_rounds, max(1, math.ceil(1.3 * min_region_ms * 1e-3 / max(probe, 1e-7)))))   # 1.3: the probe also pays one sync
    be.device_sync()
    dsdist.barrier()
    be.device_sync()
    t0 = time.perf_counter()
    wl.timing_begin()
    for _ in range(R):
        wl.run(W, K)                  # every round replays the same K resident hops; the carried state keeps evolving
    dev_ms = wl.timing_end()
    be.device_sync()
    dsdist.barrier()
    be.device_sync()
    elapsed = time.perf_counter() - t0
    wl.check(W)
    state_bytes = wl.state_bytes() if hasattr(wl, "state_bytes") else 0
    chain_min = wl.chain_min_bytes() if hasattr(wl, "chain_min_bytes") else 0
    frames_rank = B * K * T * R
    frames, t_max, ranks = dsdist.reduce_throughput(frames_rank, elapsed, device=getattr(be, "device", None))
    dev_ms = dsdist.reduce_max(dev_ms, device=getattr(be, "device", None))
    wl.close()
    launch_ms = dev_ms / (K * R)                               # average launch-to-launch duration on the kernel's stream (HIP events)
    # bytes one step moves at the least: the carried state as the library packs it, once in and once out, plus the step's samples in and out
    io = (w["M"] + 1) * w["hop"] * 4
    phys = 2.0 * state_bytes + float(B) * T * io
    basis = "state"
    if chain_min and T == 1:
        # a chain of kernels: some allocated state is shared between filters (fan form) and the stages hand spectra to each other through
        # HBM, so the step's bytes are the per-kernel budget (scripts/stage_budget.py: state touched once in and once out + stage arrays)
        phys, basis = float(chain_min), "stage_budget"
    achieved = phys / (launch_ms * 1e-3) / 1e9                 # GB/s of one GPU
    alg = algorithmic_bytes_per_frame(w, T)                    # the state counted unpacked
    survey = alg * B * T / (launch_ms * 1e-3) / 1e9
    return {
        "value": round(frames / t_max, 1), "ms_per_step": round(t_max / (K * R) * 1e3, 5), "rounds": R, "timed_steps": K * R,
        "region_ms": round(t_max * 1e3, 3), "ranks": ranks,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": w["kernel"],
                     "launches_per_step": w["launches"], "launch_ms": round(launch_ms, 5), "bytes_per_launch": phys, "bytes_basis": basis,
                     "state_bytes_per_gpu": state_bytes, "batch_per_gpu": B, "hops_per_call": T,
                     # where the carried state sits between two launches: under the 256 MiB Infinity Cache the "hbm" fraction is an on-die figure
                     "resident": "infinity_cache" if 0 < state_bytes <= 256 * 1024 * 1024 else "hbm",
                     "achieved_survey_bytes": round(survey, 1), "frac_survey_bytes": round(survey / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_frame": alg, "algorithmic_bytes_per_launch": alg * B * T},
    }


def cross_rank_verify(be, dsdist, w, T, rank, world, n_utt=8, hops=12):
    """Every rank runs the SAME small job — utterances [0, n_utt) of the workload, `hops` hops in calls of min(T, 4) hops — and the ranks
    all-gather 64-bit checksums of the enhanced samples and of the exported state: on one node they must agree bit for bit (same library, same
    kernels, same inputs; a rank on a sick GPU, a stale library on one rank or a sharding bug in the input generator shows here, where
    `isfinite` does not look).  Raises SystemExit(3) on rank 0 on a mismatch; returns {"status": "ok", "what": ...}."""
    Tc = max(1, min(int(T), 4))
    K = max(1, hops // Tc)
    wl = be.make(w, n_utt, Tc, K, 0, seed=0, graph=0)
    wl.run(0, K)
    wl.sync()
    be.device_sync()
    sums = wl.checksums() if hasattr(wl, "checksums") else [0, 0]
    wl.close()
    allv = dsdist.gather_ints(sums, device=getattr(be, "device", None))
    bad = [r for r, v in enumerate(allv) if v != allv[0]]
    if bad and rank == 0:
        sys.stderr.write("bench.py --verify: ranks %s disagree with rank 0 on utterances [0, %d): %s vs %s\n" % (bad, n_utt, [allv[r] for r in bad], allv[0]))
        sys.exit(3)
    return {"status": "ok", "what": "%d rank(s): utterances [0, %d) x %d hops on every rank, checksums of samples and exported state equal (%016x, %016x)"
                                    % (len(allv), n_utt, K * Tc, allv[0][0] & (2 ** 64 - 1), allv[0][1] & (2 ** 64 - 1))}


ACCOUNTING = {
    "state": "achieved / frac = the bytes a step must move (the carried state as the library packs it — Hermitian and symmetric matrices as "
             "triangles — once in and once out, plus the step's samples in and out) over THIS run's launch duration (HIP events on the kernel's stream)",
    "stage_budget": "achieved / frac = the per-kernel byte budget of one step (scripts/stage_budget.py: every kernel's share of the chain's state "
                    "once in and once out + the arrays its stage reads and writes; sizes from ds_chain_stage_info of this handle) over THIS run's "
                    "step duration",
    "frac_survey_bytes": "SURVEY 8(d)'s figure, which counts the same state unpacked (full complex covariance): a throughput figure in the contract's "
                         "own unit, not a physical fraction (it can exceed 1)",
    "frac_measured": "traffic = HBM bytes per step from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, "
                     "separate runs: traffic_profile); frac_measured = that / this run's launch duration / peak.  Profile-derived: tied to the build "
                     "by tests (state layout / stage budgets within 3 % of it)",
    "frac_valu_issue": "10 s-per-call regime (state moves once per 625 / 312 hops: vector-issue bound, SURVEY 8d): share of each SIMD's cycles spent "
                       "issuing the SQ_INSTS_VALU of the committed counter passes (valu_profile) at their kernels' instruction-mix cost "
                       "(scripts/kernel_mix.py) at this run's frame rate, 2.4 GHz peak clock",
}


def _profile_json(name):
    """a committed profile-derived table (profiles/<name>); a missing or malformed file is reported on stderr, never silently skipped"""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError) as e:
        sys.stderr.write("bench.py: profiles/%s unusable (%s): profile-derived figures omitted\n" % (name, e))
        return None


def attach_traffic(roof, key, warn=True):
    """HBM bytes per step from the committed rocprofv3 PMC passes of this same command (profiles/traffic_latest.json).  PMC counters cannot be
    read inside this process, so the figure is the profile's, under its own keys (`traffic`, `frac_measured`, `traffic_profile`): the live
    `achieved` / `frac` are never replaced by it."""
    t = _profile_json("traffic_latest.json")
    ent = (t or {}).get(key)
    if not ent:
        return
    roof["traffic"] = round(ent["hbm_bytes_per_launch"])
    roof["traffic_profile"] = ent.get("source", "profiles/traffic_latest.json").split(":")[0]
    gbs = ent["hbm_bytes_per_launch"] / (roof["launch_ms"] * 1e-3) / 1e9
    roof["frac_measured"] = round(gbs / HBM_PEAK_GBS, 4)
    roof["traffic_over_bytes"] = round(ent["hbm_bytes_per_launch"] / roof["bytes_per_launch"], 4)
    if warn and abs(roof["traffic_over_bytes"] - 1.0) > 0.05 and not os.environ.get("DS_BENCH_BACKEND"):
        sys.stderr.write("bench.py: %s: committed PMC traffic is %.3f x this build's byte budget — profiles/traffic_latest.json is stale for "
                         "this build or the kernel moves bytes it should not\n" % (key, roof["traffic_over_bytes"]))


def attach_compute(entry, key, frames_per_s_per_gpu):
    """The 10 s-per-call regime is bound by the vector pipes, not by HBM: `valu` = the share of every SIMD's cycles that ISSUING
    the measured vector instructions takes at this run's frame rate (profiles/compute_latest.json, committed SQ counter passes).  A separate
    object beside the live HBM `roofline`, which stays as measured."""
    t = _profile_json("compute_latest.json")
    ent = (t or {}).get(key)
    if not ent:
        return
    clk = t.get("clock_ghz_peak", 2.4) * 1e9
    vc, lc = ent["valu_issue_cycles_per_frame"], ent["lds_cycles_per_frame"]
    entry["valu"] = {"bound": "valu", "frac_valu_issue": round(frames_per_s_per_gpu * vc / clk, 4), "frac_lds": round(frames_per_s_per_gpu * lc / clk, 4),
                     "peak_ghz": round(clk / 1e9, 3), "valu_instructions_per_frame": round(ent["valu_instructions_per_frame"], 1),
                     "valu_issue_cycles_per_frame_per_simd": round(vc, 3), "lds_cycles_per_frame_per_cu": round(lc, 3),
                     "valu_profile": ent.get("profile", "?")}


def compact_roofline(r):
    # (frac_survey_bytes / bytes_basis live in the side file only since round 6: the line needed the room for tdgsc / fdgsc / host_api)
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "bytes_per_launch", "batch_per_gpu", "resident",
            "frac_measured", "traffic_profile", "value")
    return {k: r[k] for k in keep if k in r}


def compact_line(out, detail_path):
    """the ONE line the driver parses: contract keys + numeric roofline objects + per other config {value, ms_per_step, bound, frac}"""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "rounds", "timed_steps", "region_ms", "collective", "verify") if k in out}
    line["config"] = {k: v for k, v in out["config"].items() if k != "timed_hops"}      # (which hops a round replays: in the side file)
    line["roofline"] = compact_roofline(out["roofline"])
    if "valu" in out:
        line["valu"] = {k: out["valu"][k] for k in ("bound", "frac_valu_issue", "frac_lds", "valu_profile")}
    if "roofline_hbm" in out:
        line["roofline_hbm"] = compact_roofline(out["roofline_hbm"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "per_core") if k in out["cpu_baseline"]}
    oc = {}
    for name, e in out.get("other_configs", {}).items():
        c = {"value": e["value"], "ms_per_step": e["ms_per_step"]}
        if "valu" in e:
            c.update(bound="valu", frac=e["valu"]["frac_valu_issue"], frac_hbm=e["roofline"]["frac"])
            if "traffic_over_algorithmic" in e["roofline"]:
                c["traffic_x"] = e["roofline"]["traffic_over_algorithmic"]       # measured HBM bytes of a 10 s call / SURVEY 8(d)'s algorithmic bytes
        elif name in ("tdgsc", "fdgsc"):
            # one block per call through 6 / 8 dependent launches of 5 .. 35 us each that move 80 / 154 MB in all: bound by launch latency, not by
            # bytes (profiles/r06z/tdgsc_kernel_stats.csv) — `frac` is the HBM fraction of the MEASURED traffic (PMC passes), `launches` says why it is small
            c.update(bound="launch", frac=e["roofline"].get("frac_measured", e["roofline"]["frac"]), launches=e["roofline"].get("launches_per_step"))
        else:
            c.update(bound="hbm", frac=e["roofline"]["frac"])
            if "frac_measured" in e["roofline"]:
                c["frac_measured"] = e["roofline"]["frac_measured"]
        if "cpu_baseline" in e:
            c["cpu"] = e["cpu_baseline"]["value"]
        oc[name] = c
    if oc:
        line["other_configs"] = oc
    if "latency" in out:
        la = out["latency"]
        line["latency_us"] = {"chunk_ms_budget": la["chunk_ms_budget"], "pcm16_host_median": la["pcm16_host"]["median_us"], "pcm16_host_p99": la["pcm16_host"]["p99_us"],
                              "device_median": la["device"]["median_us"]}
    if "host_api" in out:
        ha = out["host_api"]
        line["host_api"] = {"unit": "[M frames/s, GB/s up + down]", "pinned_h2d_gbs": ha["pinned_h2d_gbs"]}
        for T in ("T4", "T625"):
            if T in ha:
                line["host_api"][T] = {k: [round(ha[T][k]["frames_s"] / 1e6, 2), round(ha[T][k]["gbs"], 1)] for k in ("mirror", "mirror_f32", "c_abi", "pcm16")}   # [M frames/s, GB/s both ways]
    line["detail"] = detail_path
    return line


# ------------------------------------------------------------------------------------------------
# CPU baseline leg: the oracle (a "port": plain-C double-precision restatement of the reference's loop,
# oracle/c/ds_oracle_mvdr.c, pinned to the reference's golden vectors) on the host cores, one thread per core
# ------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s=10.0):
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ds_oracle as O
    from oracle.c_oracle import COracleMVDR
    M, NFFT, HOP = 4, 512, 256
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))                   # threads actually used (reported as `cores`)
    mic = O.OracleMicArray(M=M, n_fft=NFFT)
    ang = np.array(ANGLE_DEG) / 180.0 * np.pi
    tao = O.circular_tao(mic.r, mic.c, mic.gamma, ang)
    omega = 2 * np.pi * np.arange(NFFT // 2 + 1) * FS / NFFT
    steer = np.exp(-1j * omega[:, None] * tao[None, :])
    n_distinct = 8

    def work(args):                                  # ctypes releases the GIL: threads run on separate cores
        x, = args
        eng = COracleMVDR(steer, NFFT, HOP)
        t0 = time.perf_counter()
        eng.process(x)                               # hop-by-hop inside (one hop per reference call)
        return x.shape[1] // HOP, time.perf_counter() - t0

    f0, t0 = work((O.synth_utterance(0, HOP * 200, mic),))      # calibrate on one core, then ~budget_s of work per thread
    frames = int(max(200, min(625 * 16, f0 / t0 * budget_s)))
    xs = [O.synth_utterance(u, HOP * frames, mic) for u in range(n_distinct)]   # inputs built before the timed region
    t_start = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        res = list(pool.map(work, [(xs[u % n_distinct],) for u in range(cores)]))
    wall = time.perf_counter() - t_start
    busy = max(r[1] for r in res)
    total = sum(r[0] for r in res)
    # one 1024-sample chunk (4 hops) of ONE stream on ONE core: the C port's figure beside the `latency` block
    eng = COracleMVDR(steer, NFFT, HOP)
    xc = O.synth_utterance(99, 1024 * 300, mic)
    tc = []
    for i in range(300):
        t1 = time.perf_counter()
        eng.process(xc[:, i * 1024:(i + 1) * 1024])
        tc.append(time.perf_counter() - t1)
    return {"value": round(total / busy, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d streams x %d hops (one hop per call; %d distinct synthetic utterances), oracle/c/ds_oracle_mvdr.c "
                      "(plain C, fp64), one thread per core; %.1f s wall" % (cores, frames, n_distinct, wall),
            "per_core": round(total / busy / cores, 1),
            "chunk_1024_one_core_us": {"median_us": round(float(np.median(tc)) * 1e6, 1), "p99_us": round(float(np.percentile(tc, 99)) * 1e6, 1)}}


# CPU baseline of the chain / GSC configs: the NumPy restatements of oracle/ds_oracle.py (fp64, pinned to the reference's golden vectors),
# one worker process per host core, each on its own synthetic utterance, one hop per call like the reference's loop
def _cpu_oracle_worker(args):
    name, seed, frames = args
    import numpy as np
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import ds_oracle as O
    w = WORKLOADS[name]
    M, nfft, hop = w["M"], w["nfft"], w["hop"]
    mic = O.OracleMicArray(arrayType="circular", r=w["r"], M=M, n_fft=nfft)
    ang = np.array(ANGLE_DEG) / 180.0 * np.pi
    x = O.synth_utterance(seed, hop * frames, mic) * (0.2 if name != "cfg4" else 1.0)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        if name == "cfg3":
            O.OracleGSC(mic, nfft, with_dead_state=False).process(x, ang, 2)
        elif name == "mvdr_pf":
            O.OracleMvdrPostfilter(mic, nfft=nfft, hop=hop).process(x, ang, 2)
        elif name == "cfg4":
            O.OracleWpeMvdrPostfilter(mic, nfft=nfft, hop=hop).process(x, ang)
        else:
            O.OracleSubbandGSC(mic, frameLen=hop, rls_bm=True).process(x)
    return frames, time.perf_counter() - t0


def cpu_baseline_oracle(name, budget_s=3.0):
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    f0, t0 = _cpu_oracle_worker((name, 0, 24))                      # calibrate on one core (includes the first-frame branches)
    frames = int(max(24, min(2000, f0 / t0 * budget_s)))
    t_start = time.perf_counter()
    with ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn")) as pool:
        res = list(pool.map(_cpu_oracle_worker, [(name, 1 + u, frames) for u in range(cores)]))
    wall = time.perf_counter() - t_start
    busy = max(r[1] for r in res)
    total = sum(r[0] for r in res)
    return {"value": round(total / busy, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d processes x %d hops (one synthetic utterance each), oracle/ds_oracle.py (NumPy, fp64), one process per core; %.1f s wall incl. start-up"
                      % (cores, frames, wall),
            "per_core": round(total / busy / cores, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=625)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--config", default="cfg2", choices=sorted(WORKLOADS), help="BASELINE config measured as the headline (default cfg2)")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: the config's BASELINE batch)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling: this many utterances in all, sharded contiguously over the ranks "
                                                               "(dist.shard_range), e.g. cfg4's 8192 over 8 GPUs; default 0 = weak scaling")
    ap.add_argument("--hops-per-step", type=int, default=1, help="T: hops per call (1 = streaming callback regime)")
    ap.add_argument("--graph", type=int, default=-1, help="1 = replay a round as one hipGraph, 0 = plain launches (default: per config)")
    ap.add_argument("--min-region-ms", type=float, default=250.0, help="rounds of --steps are repeated until the timed region is this long")
    ap.add_argument("--hbm-batch", type=int, default=16384, help="batch of the roofline_hbm regime (state working set > the 256 MiB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_hbm and other_configs (profiling runs)")
    ap.add_argument("--host-api", action="store_true", help="only the host-buffer API entry (adaptivebeamfomer.process(x_numpy) at batch): prints its JSON")
    ap.add_argument("--verify", action="store_true", help="every rank also runs utterances [0, 8) of the job for 12 hops; the ranks all-gather 64-bit "
                                                          "checksums of the enhanced samples and of the exported state, rank 0 asserts that they agree "
                                                          "(the line then carries \"verify\": \"ok\"); on by default when --gpus > 1")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("--gpus and --steps must be >= 1, --warmup >= 0")

    profiled = under_profiler()
    hw_queues_raised = False
    if profiled:
        # the profiler's library brought the runtime up before main(): only a limit that was in the environment BEFORE the profiler started
        # is the one the runtime has (scripts/profile_bench.sh exports it, so that the traced run is the pipelined chain the bench measures)
        try:
            hw_queues_raised = int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 6
        except ValueError:
            hw_queues_raised = False
    if not profiled:
        # this PROCESS is the application: hardware queues per device for the HIP runtime (default 4) — the chain handles run their stages on
        # up to five streams and two streams that share a queue serialise.  Must be in the environment before the first HIP call (the child
        # ranks inherit it); under a profiler the runtime is already up and the variable is left alone (DS_PARAM_TAIL_ASYNC follows it)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        try:
            hw_queues_raised = int(os.environ["GPU_MAX_HW_QUEUES"]) >= 6      # what THIS process (or the parent that launched the ranks) asked the runtime for
        except ValueError:
            hw_queues_raised = False
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if profiled:
            sys.stderr.write("bench.py: --gpus %d under a GPU profiler refused: the profiled process has initialised the GPU and must not start "
                             "child ranks (profile one GPU: scripts/profile_bench.sh)\n" % args.gpus)
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))        # children only; nothing in this process has touched torch or the GPU

    from distantspeech_amd import dist as dsdist
    rank, local_rank, world = dsdist.env_world()
    # this rank's launch thread on the cores of its GPU's NUMA node — before the first GPU call of the process (sysfs only)
    affinity = dsdist.pin_to_gpu_numa_node(int(os.environ.get("DS_FORCE_DEVICE", local_rank))) if not os.environ.get("DS_BENCH_BACKEND") else "off"
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but %d rank(s) were launched (WORLD_SIZE)\n" % (args.gpus, world))
        sys.exit(2)
    # the NumPy-oracle CPU baselines of the other configs run in worker processes: started here, before this process touches the GPU
    cpu_pre = {}
    if world == 1 and not args.no_cpu_baseline and not args.no_extras and args.config == "cfg2" and not os.environ.get("DS_BENCH_BACKEND") \
            and not profiled:                            # worker processes: never from a profiled (GPU-initialised) process
        for name in ("mvdr_pf", "cfg3", "cfg4", "cfg5"):
            cpu_pre[name] = cpu_baseline_oracle(name)
    be = load_backend(local_rank, world, tail_async=hw_queues_raised)
    dsdist.init(backend=be.dist_backend)

    if args.host_api:
        if rank == 0:
            print(json.dumps(host_api_entry(be)), flush=True)
        dsdist.finalize()
        return
    w = WORKLOADS[args.config]
    B = args.batch or batch_of(w, args.hops_per_step)
    seed = None
    if args.total_batch:
        lo, hi = dsdist.shard_range(args.total_batch, rank, world)     # utterances [lo, hi) of the job live on this rank's GPU
        B, seed = hi - lo, lo
        if B < 1:
            raise SystemExit("--total-batch %d leaves rank %d without an utterance" % (args.total_batch, rank))
    K, W, T = args.steps, args.warmup, args.hops_per_step
    graph = None if args.graph < 0 else args.graph
    res = measure(be, dsdist, w, B, T, K, W, rank, world, args.min_region_ms, graph=graph, seed=seed)
    if res["ranks"] != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but %d rank(s) contributed to the result\n" % (args.gpus, res["ranks"]))
        sys.exit(2)
    verify = None
    if args.verify or world > 1:
        verify = cross_rank_verify(be, dsdist, w, T, rank, world)
    out = None
    if rank == 0:
        regime = "streaming callback regime" if T == 1 else "chunked"
        mics = {"cfg2": "4-mic, 512-FFT", "cfg3": "4-mic, 512-FFT", "fixed": "4-mic, 512-FFT", "cfg4": "8-mic, 1024-FFT", "cfg5": "6-mic, 512 bands", "wpe_nb": "4-ch WPE, 256 bands", "cfg4_n10": "8-mic, 1024-FFT, 10-tap WPE", "mvdr_pf": "4-mic, 512-FFT, MVDR + post-filter", "nb_mvdr": "6-mic, 512-FFT, notebook online MVDR", "nb_mvdr_m4": "4-mic, 512-FFT, notebook online MVDR",
                "tdgsc": "4-mic, block 256", "fdgsc": "4-mic, block 256"}[args.config]
        out = {
            "metric": "enhanced frames/sec (%s)" % mics, "value": res["value"], "unit": "frames/s",
            "n_gpus": res["ranks"], "steps": K, "warmup": W, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak" if not args.total_batch else "strong", "vs_baseline": None, "dtype": "f32", "data": be.name,
            "rounds": res["rounds"], "timed_steps": res["timed_steps"], "region_ms": res["region_ms"],
            "collective": dsdist.collective_name(),       # what the barrier and the final reduce ran over: "nccl" (= RCCL), "gloo", or "none" (one process)
            "config": {"workload": "%s: %s, batch=%d per GPU, %d hop(s) per call (%s)"
                                   % (("BASELINE " + args.config) if args.config.startswith("cfg") else args.config, w["desc"], B, T, regime),
                       "batch_per_gpu": B, "hops_per_call": T, "n_mics": w["M"], "nfft": w["nfft"], "hop": w["hop"],
                       "launch": "hipGraph" if (w["graph"] if graph is None else graph) else "plain",
                       "timed_hops": "each round replays hops %d..%d of every utterance (the recipe's source is on until hop %d, then off for 0.5 s); "
                                     "state keeps evolving" % (W * T + (LEAD_HOPS if T == 1 else 0), (W + K) * T - 1 + (LEAD_HOPS if T == 1 else 0),
                                                               int(0.5 * FS / w["hop"]))},
            "roofline": res["roofline"],
        }
        if verify is not None:
            out["verify"] = verify["status"]
            out["config"]["verify"] = verify["what"]
        out["config"]["affinity"] = affinity
        if args.total_batch:
            out["config"]["total_batch"] = args.total_batch
        if getattr(be, "shared_device", False):
            out["config"]["note"] = "all %d ranks share GPU %d (DS_FORCE_DEVICE, gloo): rank-path check, not a scaling measurement" % (world, be.local_rank)
        if T == 1 and B == batch_of(w, 1):
            attach_traffic(out["roofline"], args.config)
        if T > 1:
            attach_compute(out, "%s_10s_chunks" % args.config if T >= 300 else "none", out["value"] / max(1, res["ranks"]))

    detail = {}
    if not args.no_extras:
        # (1) the same kernel with the state working set outside the Infinity Cache: an HBM measurement of the HBM claim
        if args.config in ("cfg2", "cfg3", "fixed") and T == 1 and args.hbm_batch > B and not args.total_batch:
            Kh, Wh = min(K, 40), min(W, 5)
            # one launch per step (split = 1), so that the launch duration is the kernel's: the library's default at this batch is two
            # utterance groups on two streams (+3 % here)
            r2 = measure(be, dsdist, w, args.hbm_batch, 1, Kh, Wh, rank, world, min(args.min_region_ms, 150.0), graph=graph, split=1)
            if rank == 0:
                roof = r2["roofline"]
                roof.update(value=r2["value"], steps=Kh, rounds=r2["rounds"], state_bytes_per_launch=w["S"] * args.hbm_batch,
                            note="same kernel, batch %d per GPU: the carried state of one launch exceeds the 256 MiB Infinity Cache, so consecutive "
                                 "launches are served by HBM" % args.hbm_batch)
                attach_traffic(roof, args.config + "_hbm")
                out["roofline_hbm"] = roof
        # (2) the other BASELINE configs through the same path (same sharding, same bracketing)
        others = {}
        for name in EXTRA_T1:
            if name == args.config or T != 1 or args.total_batch:
                continue
            wo = WORKLOADS[name]
            Ko, Wo = min(K, 40), min(W, 4)
            ro = measure(be, dsdist, wo, wo["batch"], 1, Ko, max(Wo, 2), rank, world, min(args.min_region_ms, 120.0))
            if rank == 0:
                # (the two overlap-save chains: their handles carry state a block does not touch — the post-filter's, the two-path foreground
                # filters' — so the state-payload budget overstates their bytes; their fraction is the MEASURED one, below)
                # (the notebook operator: 204 B of scratch per lane at 6 microphones = 1.28 x its byte budget on the counters; at 4096
                # utterances part of the 4-microphone state stays in the Infinity Cache = 0.95 x — both known, neither a stale profile)
                attach_traffic(ro["roofline"], name, warn=name not in ("tdgsc", "fdgsc", "nb_mvdr", "nb_mvdr_m4"))
                others[name] = {"workload": "%s: %s, batch=%d per GPU, 1 hop per call" % (name, wo["desc"], wo["batch"]),
                                "value": ro["value"], "unit": "frames/s", "n_gpus": ro["ranks"], "steps": Ko, "rounds": ro["rounds"],
                                "ms_per_step": ro["ms_per_step"], "roofline": ro["roofline"]}
                if name in ("nb_mvdr", "nb_mvdr_m4"):
                    # the notebook's online MVDR is bound by arithmetic at ANY call length (the estimation core and the principal eigenvector in
                    # double per bin and frame: 16.7 k vector instructions per frame against 0.3 MB of traffic per step at 6 microphones; SQ passes:
                    # SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES = 0.86): the per-frame instruction count is the 10 s profile's, the vector-issue
                    # fraction follows from this run's frame rate
                    attach_compute(others[name], name + "_10s_chunks", ro["value"] / max(1, ro["ranks"]))
        # (3) the BASELINE configs as SURVEY 8(d) words their inputs: 10 s per utterance in ONE call (625 hops at hop 256, 312 at hop 512;
        # cfg5's "10 s streaming chunks").  The carried state then moves once per chunk and the step is bound by the per-hop arithmetic /
        # LDS work of the kernels — the HBM fraction is reported for completeness, not as the limiter
        if args.config == "cfg2" and T == 1 and not args.total_batch:
            for name, Tc in EXTRA_CHUNKED:
                wo = WORKLOADS[name]
                Bc = batch_of(wo, Tc)
                ro = measure(be, dsdist, wo, Bc, Tc, 2, 1, rank, world, min(args.min_region_ms, 120.0))
                if rank == 0:
                    ent = {"workload": "%s: %s, batch=%d per GPU, 10 s per call (%d hops)" % (name, wo["desc"], Bc, Tc),
                           "value": ro["value"], "unit": "frames/s", "n_gpus": ro["ranks"], "steps": 2, "rounds": ro["rounds"],
                           "ms_per_step": ro["ms_per_step"], "hops_per_call": Tc, "roofline": ro["roofline"]}
                    attach_compute(ent, name + "_10s_chunks", ro["value"] / max(1, ro["ranks"]))
                    attach_traffic(ent["roofline"], name + "_10s_chunks", warn=False)      # HBM bytes a 10 s call really moves (committed PMC passes)
                    if ent["roofline"].get("traffic"):
                        alg_step = algorithmic_bytes_per_frame(wo, Tc) * Bc * Tc
                        ent["roofline"]["traffic_over_algorithmic"] = round(ent["roofline"]["traffic"] / alg_step, 3)
                    others[name + "_10s_chunks"] = ent
        if rank == 0 and others:
            out["other_configs"] = others

    if rank == 0 and world == 1 and not args.no_extras and args.config == "cfg2" and hasattr(be, "latency"):
        out["latency"] = be.latency()
        out["host_api"] = host_api_entry(be)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2" and not os.environ.get("DS_BENCH_BACKEND"):
            out["cpu_baseline"] = cpu_baseline()
            if "latency" in out:
                out["latency"]["cpu_port_one_core"] = out["cpu_baseline"]["chunk_1024_one_core_us"]
            for name in out.get("other_configs", {}):
                if name in cpu_pre:
                    out["other_configs"][name]["cpu_baseline"] = cpu_pre[name]
        # the side file: everything, with the prose; the line: numbers only (the driver keeps an 8 KB tail of stdout)
        detail_path = os.environ.get("DS_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        out["accounting"] = ACCOUNTING
        out["data_note"] = DATA_NOTE
        try:
            with open(detail_path, "w") as fh:
                json.dump(out, fh, indent=1)
            shown = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s (%s)\n" % (detail_path, e))
            shown = None
        line = json.dumps(compact_line(out, shown), separators=(",", ":"))
        assert len(line) <= 4096, "bench line is %d bytes (> 4 KB): the driver's tail would cut it" % len(line)
        print(line, flush=True)
    dsdist.finalize()


if __name__ == "__main__":
    main()
