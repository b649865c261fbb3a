#!/usr/bin/env python3
"""File-level driver equal to the reference's example/run_MVDRbeamformer.py:23-61 (BASELINE config 2 on one recording): load a
directory of single-channel WAVs, run the MCRA-gated adaptive MVDR beamformer towards 197 degrees, optionally save the result.

    python examples/run_MVDRbeamformer.py --input DIR [--save out.wav] [--method 2] [--angle 197] [--frame 512]
(the reference's script hard-codes test_audio/rec1 and a 256-point frame; adaptivebeamfomer.process only works one hop per call at
HEAD — here a call with T hops is T one-hop calls, so the whole file goes through in one call).  method: 0 src, 1 DS, 2 MVDR, 3 TFGSC."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distantspeech_amd import MicArray, adaptivebeamfomer           # noqa: E402
from distantspeech_amd.utils import load_wav, save_audio             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True, help="directory with one .wav per microphone")
    ap.add_argument("--save", default=None, help="output wav path")
    ap.add_argument("--method", type=int, default=2)
    ap.add_argument("--angle", type=float, default=197.0, help="look direction, degrees")
    ap.add_argument("--frame", type=int, default=512, help="frame length = FFT size (hop = frame / 2)")
    args = ap.parse_args()
    x, sr = load_wav(os.path.abspath(args.input))                      # [ch, samples]
    hop = args.frame // 2
    x = x[:, : (x.shape[1] // hop) * hop]
    mic = MicArray(arrayType='circular', r=0.032, M=x.shape[0], n_fft=args.frame)
    bf = adaptivebeamfomer(mic, args.frame, hop, args.frame, 343, 0.032, sr)
    angle = np.array([args.angle, 0]) / 180 * np.pi
    t0 = time.perf_counter()
    yout = bf.process(x, angle, method=args.method)
    dt = time.perf_counter() - t0
    print("%d channels x %.1f s processed in %.3f s (%.0fx real time)" % (x.shape[0], x.shape[1] / sr, dt, x.shape[1] / sr / dt))
    if args.save:
        save_audio(args.save, yout['data'], fs=sr)
    return yout['data']


if __name__ == "__main__":
    main()
