#!/usr/bin/env python3
"""Byte budget per launch of the two chain configurations at one block per call: for every kernel of a cfg4 / cfg5 step,
the bytes it MUST move (its share of the carried state once in and once out + the stage's input and output arrays) beside the HBM bytes the
PMC passes measured (traffic.json of scripts/profile_bench.sh).

    python scripts/stage_budget.py <cfg4|cfg5> <traffic.json> [--batch B] > profiles/<round>/<cfg>_stage_budget.md

Runs ON THE GPU BOX: the state sizes of the stages are read from the library (ds_chain_stage_info on a handle of the bench's shape), not
retyped here.  What a kernel touches of a stage's state is stated per row (some stages keep planes a given kernel never reads: the fan form
of the blocking filters maintains the tap buffer and P of the first filter only; McCDR's rows of the McSpp state belong to the analysis
kernel).  State planes are float4 planes [b][f/4][k][4] (ds_ops.hpp st_index): a lane moves 16 B per plane, K lanes per utterance."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def planes(nf):
    return (nf + 3) // 4


def budget_rows(cfg, w, stages, B):
    """(kernel substring, stage label, what it touches, state bytes per utterance one way, in bytes, out bytes) per kernel of one step of the
    chain `cfg` (bench.WORKLOADS key), from the stage table the library reports (BatchEngine.chain_stages())."""
    M, K, hop, N = w["M"], w["nfft"] // 2 + 1, w["hop"], w.get("filter_len", 2)
    P16 = K * 16                       # one float4 plane of one utterance as the K lanes move it
    rows = []
    if cfg == "cfg5":
        # taps of the alignment FIR bank the bench hands the chain (bench.py GpuWorkload: fractional_delay_filter_bank of the look direction)
        import numpy as np
        from distantspeech_amd.mic_array import MicArray, compute_tau
        from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
        mic = MicArray(arrayType="circular", r=w["r"], M=M, n_fft=w["nfft"])
        tau = compute_tau(mic, np.array(bench.ANGLE_DEG) / 180.0 * np.pi)
        Lt = fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic.fs).shape[0]
        front3 = [
            ("ds_dcnotch_kernel", "DC notch (stage 0)", "notch memory (2 floats per channel)", M * 2 * 4, M * hop * 4, M * hop * 4),
            ("ds_fir_kernel", "FIR bank + channel mean (stage 0)", "FIR history, %d samples per channel" % (Lt - 1), (Lt - 1) * M * 4, M * hop * 4, M * hop * 4 + hop * 4),
            ("ds_stft_cdr_kernel", "analysis of M channels + McCDR (stages 1, 2)", "analysis overlap M x hop; McCDR rows 0..8 of the McSpp state = 3 planes",
             M * hop * 4 + 3 * P16, M * hop * 4, K * M * 8 + K * 4 + 4),
        ]
        # round 4's shelved experiment: the three as ONE kernel (ds_front_kernel; make SHELVED=1 + DS_CHAIN_FRONT_FUSED=1): the notched and
        # the aligned channels never leave the workgroup
        front1 = [
            ("ds_front_kernel", "DC notch + FIR bank + channel mean + analysis of M channels + McCDR (stages 0, 1, 2)",
             "notch memory, FIR history (%d samples per channel), analysis overlap M x hop, McCDR rows 0..8 of the McSpp state = 3 planes" % (Lt - 1),
             M * 2 * 4 + (Lt - 1) * M * 4 + M * hop * 4 + 3 * P16, M * hop * 4, K * M * 8 + K * 4 + 4 + hop * 4),
        ]
        mcspp_planes = planes(12 + 2 * M * M + 3) - 3
        if os.environ.get("DS_CHAIN_FAN_FUSED") != "1" or w.get("rls_lambda", 0.0) <= 0.0:
            middle = [
                ("ds_binop_kernel<13", "McSpp, steady-state build (stage 2)", "rows 12.. of the McSpp state: %d planes" % mcspp_planes,
                 mcspp_planes * P16, K * M * 8 + K * 4 + 4, K * 4),
                ("ds_stft_rows_kernel", "analysis of the fixed beamformer output (stage 3)", "analysis overlap, one channel", hop * 4, hop * 4, K * 8),
                ("ds_subrls_fan_kernel", "M RLS blocking filters, fan form (stage 5)", "filter 0: W, X, P = 4 planes; filters 1..M-1: W = 1 plane each",
                 (4 + (M - 1)) * P16, K * 8 + K * M * 8, M * K * 8),
            ]
        else:
            # round 5's shelved experiment (make SHELVED=1 + DS_CHAIN_FAN_FUSED=1): the RLS filters inside McSpp's launch — the frame's spectra read once for both
            middle = [
                ("ds_stft_rows_kernel", "analysis of the fixed beamformer output (stage 3)", "analysis overlap, one channel", hop * 4, hop * 4, K * 8),
                ("ds_binop_kernel<18", "McSpp, steady-state build, with the M RLS blocking filters in the same threads (stages 2, 5)",
                 "rows 12.. of the McSpp state: %d planes; blocking filter 0: W, X, P = 4 planes; filters 1..M-1: W = 1 plane each" % mcspp_planes,
                 (mcspp_planes + 4 + (M - 1)) * P16, K * M * 8 + K * 4 + 4 + K * 8, K * 4 + M * K * 8),
            ]
        rows = (front1 if os.environ.get("DS_CHAIN_FRONT_FUSED") == "1" else front3) + middle + [
            ("ds_frames_kernel", "tail: synthesis of the M blocking outputs, re-analysis, canceller, synthesis (stages 4, 6, 7, 8)",
             "synthesis + analysis overlaps 2 x M x hop, output overlap hop, delayed fixed spectrum K x 8, canceller state %d planes" % planes(4 * N * M + 1),
             2 * M * hop * 4 + hop * 4 + K * 8 + planes(4 * N * M + 1) * P16, M * K * 8 + K * 8 + K * 4, hop * 4),
        ]
    elif cfg == "cfg4":
        # an operator handle's payload also holds the analysis / synthesis overlaps every handle allocates (M + 1 rows of nfft - hop floats);
        # the operator kernels never touch them
        idle = (M + 1) * (w["nfft"] - hop) * 4
        wpe_state = stages[1]["bytes"] // B - idle
        mcmcra_state = stages[2]["bytes"] // B - idle
        mvdr_state = stages[3]["bytes"] // B - idle
        rows = [
            ("ds_stft_kernel", "analysis of M channels (stage 0)", "analysis overlap M x hop", M * hop * 4, M * hop * 4, K * M * 8),
            ("ds_wpe_kernel", "WPE, %d taps (stage 1)" % N, "the whole WPE state (inverse covariance as a packed triangle, taps, tap buffer) + one frame in and "
             "one frame out of the delay line", wpe_state + K * M * 8, K * M * 8, K * M * 8),
            ("ds_binop_kernel<1,", "McMcra speech presence (stage 2)", "the whole McMcra state, %d planes" % planes(M * (M + 1) + 4),
             mcmcra_state, K * M * 8, K * 4 + K * 4),
            ("ds_binop_kernel<11", "adaptive MVDR + SPP gain on frames (stage 3)", "the whole covariance / MCRA state", mvdr_state, K * M * 8 + K * 4 + K * 4, K * 8),
            ("ds_istft_rows_kernel", "synthesis (stage 4)", "synthesis overlap, one channel", hop * 4, K * 8, hop * 4),
        ]
    return rows


def minimal_step_bytes(cfg, w, eng, B):
    """bytes ONE step (one block of every utterance) of the chain must move: every kernel's share of the state once in and once out + the
    arrays its stage reads and writes (bench.py quotes it as the chains' live `roofline.achieved`)."""
    stages = {i: dict(algo=a, mics=m, batch=b, bytes=n) for i, a, m, b, n in eng.chain_stages()}
    return float(sum(B * (2 * st + bi + bo) for _, _, _, st, bi, bo in budget_rows(cfg, w, stages, B)))


def main():
    cfg, traffic_path = sys.argv[1], sys.argv[2]
    w = bench.WORKLOADS[cfg]
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else w["batch"]
    from distantspeech_amd import BatchEngine, _lib as L
    eng = BatchEngine(getattr(L, "ALGO_" + w["algo"]), w["M"], w["nfft"], w["hop"], batch=B, filter_len=w.get("filter_len", 0),
                      rls_lambda=w.get("rls_lambda", 0.0))
    stages = {i: dict(algo=a, mics=m, batch=b, bytes=n) for i, a, m, b, n in eng.chain_stages()}
    total_state = eng.state_bytes()
    parts = int(os.environ.get("DS_CHAIN_PARTS", "2" if (cfg == "cfg4" and B >= 512) else "1"))
    eng.close()
    M, K, hop, N = w["M"], w["nfft"] // 2 + 1, w["hop"], w.get("filter_len", 2)
    rows = budget_rows(cfg, w, stages, B)
    tr = json.load(open(traffic_path))
    print("# %s at one block per call: bytes per launch, minimal against measured" % cfg)
    print()
    print("`%s`, B = %d utterances per GPU%s.  Carried state of the whole chain as the library packs it: %.1f MB (%d B per utterance)."
          % (w["desc"], B, ", %d utterance groups (each kernel is launched once per group; rows are per launch)" % parts if parts > 1 else "",
             total_state / 1e6, total_state // B))
    print("Measured = HBM bytes of the PMC passes (`%s`, FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, mean over the launches of the run)."
          % os.path.relpath(traffic_path, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    print("Minimal = the kernel's share of the state once in and once out + the stage's input and output arrays; K = %d lanes of a %d-lane plane "
          "row are live." % (K, (K + 7) & ~7))
    print()
    print("| kernel | stage | state the kernel touches | state B/utt (one way) | in B/utt | out B/utt | minimal MB/launch | measured MB/launch | measured / minimal |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_min = tot_meas = 0.0
    out = []
    for sub, label, what, st, bi, bo in rows:
        ks = [k for k in tr["kernels"] if sub in k]
        if not ks:
            print("| `%s` | %s | not in the trace | | | | | | |" % (sub, label))
            continue
        k = ks[0]
        meas = tr["kernels"][k]["hbm_bytes_per_launch"]
        Bl = B / parts
        mn = Bl * (2 * st + bi + bo)
        launches_per_step = parts
        tot_min += mn * launches_per_step
        tot_meas += meas * launches_per_step
        out.append(dict(kernel=k, stage=label, touches=what, state_bytes_per_utt=st, in_bytes_per_utt=bi, out_bytes_per_utt=bo, minimal_bytes_per_launch=mn,
                        measured_bytes_per_launch=meas, ratio=meas / mn))
        print("| `%s` | %s | %s | %d | %d | %d | %.1f | %.1f | %.2f |" % (k.replace("void ds::", "").split("(")[0], label, what, st, bi, bo, mn / 1e6, meas / 1e6, meas / mn))
    print("| **step** | | | | | | **%.1f** | **%.1f** | **%.2f** |" % (tot_min / 1e6, tot_meas / 1e6, tot_meas / tot_min))
    print()
    print("<!-- json: %s -->" % json.dumps(dict(config=cfg, batch=B, groups=parts, rows=out, minimal_bytes_per_step=tot_min, measured_bytes_per_step=tot_meas)))


if __name__ == "__main__":
    main()
