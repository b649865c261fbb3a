#!/usr/bin/env python3
"""File-level driver equal to the reference's example/run_fixedbeamformer.py:23-73 (BASELINE config 1): load a directory of
single-channel WAVs, run the fixed beamformer (delay-and-sum or superdirective) towards 197 degrees, optionally save the result.

    python examples/run_fixedbeamformer.py --input DIR [--save out.wav] [--weights DS|SD] [--angle 197] [--frame 512]
(the reference's script hard-codes test_audio/rec1, a 256-point frame and a process() signature — angle, method, retH, retWNG, retDI —
that FixedBeamformer.process no longer has at HEAD, fixedbeamformer.py:167; this one takes its inputs as arguments and calls the class
as it stands: FixedBeamformer(mic, frameLen, hop, nfft, c, fs).process(x[samples, channels], angle))."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distantspeech_amd import FixedBeamformer, MicArray           # noqa: E402
from distantspeech_amd.utils import load_wav, save_audio           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True, help="directory with one .wav per microphone")
    ap.add_argument("--save", default=None, help="output wav path")
    ap.add_argument("--weights", default="DS", choices=["DS", "SD"], help="delay-and-sum or superdirective weights")
    ap.add_argument("--angle", type=float, default=197.0, help="look direction, degrees")
    ap.add_argument("--frame", type=int, default=512, help="frame length = FFT size (hop = frame / 2)")
    args = ap.parse_args()
    x, sr = load_wav(os.path.abspath(args.input))                      # [ch, samples]
    hop = args.frame // 2
    x = x[:, : (x.shape[1] // hop) * hop]
    mic = MicArray(arrayType='circular', r=0.032, M=x.shape[0], n_fft=args.frame)
    fb = FixedBeamformer(mic, args.frame, hop, args.frame, 343, sr, weightType=args.weights)
    t0 = time.perf_counter()
    y = fb.process(x.T, (args.angle, 0))                               # degrees, like the class's own default [197, 0]
    dt = time.perf_counter() - t0
    print("%d channels x %.1f s processed in %.3f s (%.0fx real time)" % (x.shape[0], x.shape[1] / sr, dt, x.shape[1] / sr / dt))
    if args.save:
        save_audio(args.save, y, fs=sr)
    return y


if __name__ == "__main__":
    main()
