"""debug: cfg4 chain on a period-3 frame sequence: which stage's state goes non-finite first"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from distantspeech_amd import _lib as L
be = bench.GpuBackend(0, 1)
w = bench.WORKLOADS["cfg4"]
wl = be.make(w, 48, 1, 3, 1, seed=0, graph=0)
wl.run(0, 1)
for r in range(200):
    wl.run(1, 3); wl.sync()
    yfin = bool(torch.isfinite(wl.y).all())
    sts = {i: wl.eng.stage_state_raw(i) for i in (1, 2, 3)}
    fin = {i: bool(np.isfinite(s).all()) for i, s in sts.items()}
    mx = {i: float(np.nanmax(np.abs(s))) for i, s in sts.items()}
    if r % 10 == 0 or not yfin or not all(fin.values()):
        print(r, "y finite", yfin, "state finite", fin, "absmax", {k: "%.3e" % v for k, v in mx.items()}, "y rms %.3e" % float(wl.y[:, 512:].pow(2).mean().sqrt()))
    if not yfin or not all(fin.values()):
        for i, s in sts.items():
            if not fin[i]:
                bad = np.argwhere(~np.isfinite(s))
                print(" stage", i, "bad utterances", sorted(set(bad[:, 0].tolist()))[:8], "first flat idx", bad[0], "per-utt floats", s.shape[1])
        break
