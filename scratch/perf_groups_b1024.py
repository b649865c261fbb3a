"""cfg2 at its BASELINE batch (1024 utterances) as 1, 2 and 4 free-running utterance groups (DS_PARAM_SPLIT) through bench.measure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from distantspeech_amd import dist as dsdist
be = bench.load_backend(0, 1)
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
for rep in range(3):
    for split in (1, 2, 4):
        r = bench.measure(be, dsdist, w, w["batch"], 1, 20, 5, 0, 1, 150.0, split=split)
        print("split %d: %.2f M frames/s  %.2f us per step" % (split, r["value"] / 1e6, r["ms_per_step"] * 1e3), flush=True)
