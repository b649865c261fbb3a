#!/usr/bin/env python3
"""File-level driver equal to the reference's example/run_GSC.py:22-85 (offline branch): load a directory of
single-channel WAVs, run the GSC beamformer towards 197 degrees, optionally save the result.

    python examples/run_GSC.py --input DIR [--save out.wav] [--method 2] [--angle 197]
(the reference's script hard-codes test_audio/rec1 and calls GSC with a constructor signature that no longer exists;
this one takes the directory as an argument and uses the class as it stands at HEAD: GSC(mic, frameLen, angle))."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distantspeech_amd import GSC, MicArray           # noqa: E402
from distantspeech_amd.utils import load_wav, save_audio   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True, help="directory with one .wav per microphone")
    ap.add_argument("--save", default=None, help="output wav path")
    ap.add_argument("--method", type=int, default=2)
    ap.add_argument("--angle", type=float, default=197.0)
    args = ap.parse_args()
    x, sr = load_wav(os.path.abspath(args.input))                      # [ch, samples]
    frameLen = 512
    hop = frameLen // 2
    x = x[:, : (x.shape[1] // hop) * hop]
    mic = MicArray(arrayType='circular', r=0.032, M=x.shape[0], n_fft=frameLen)
    gsc = GSC(mic, frameLen, angle=[args.angle, 0])
    angle = np.array([args.angle, 0]) / 180 * np.pi
    t0 = time.perf_counter()
    yout = gsc.process(x, angle, method=args.method)
    dt = time.perf_counter() - t0
    print("%d channels x %.1f s processed in %.3f s (%.0fx real time)" % (x.shape[0], x.shape[1] / sr, dt, x.shape[1] / sr / dt))
    if args.save:
        save_audio(args.save, yout['data'], fs=sr)
    return yout['data']


if __name__ == "__main__":
    main()
