"""debug: cfg4 chain, B = 48, bench synth with seed 48 (rank 1 of --total-batch 96 --gpus 2): where does a non-finite value appear?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from distantspeech_amd import _lib as L
be = bench.GpuBackend(0, 1)
w = bench.WORKLOADS["cfg4"]
for seed in (0, 48):
    for graph in (0, 1):
        wl = be.make(w, 48, 1, 3, 1, seed=seed, graph=graph)
        x = wl.x
        print("seed", seed, "graph", graph, "x finite", bool(torch.isfinite(x).all()), "x absmax", float(x.abs().max()))
        wl.run(0, 1); wl.run(1, 3); wl.sync(); wl.run(1, 3); wl.sync()
        nbad = -1
        for r in range(400):
            wl.run(1, 3)
            if r % 20 == 19:
                wl.sync()
                if not bool(torch.isfinite(wl.y).all()):
                    nbad = r; break
        wl.sync()
        print('   first bad round', nbad)
        y = wl.y
        bad = ~torch.isfinite(y)
        print("   y nonfinite count", int(bad.sum()), "utterances", torch.nonzero(bad.any(dim=1)).flatten().tolist()[:10],
              "first sample idx", (torch.nonzero(bad)[0].tolist() if bad.any() else None))
        wl.close()
