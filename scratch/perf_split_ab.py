"""A/B: one chain handle of B utterances against S handles of B/S utterances on S streams (sub-batch pipelining).
   usage: python scratch/perf_split_ab.py cfg5 [K]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = bench.WORKLOADS[name]
be = bench.load_backend(0, 1)
B = int(os.environ.get("BATCH", w["batch"]))
for S in [int(v) for v in os.environ.get('SPLITS', '1,2,4,8,1').split(',')]:
    wls = [be.make(w, B // S, 1, K, 5, seed=s, graph=w["graph"]) for s in range(S)]
    for wl in wls: wl.run(0, 5)
    for wl in wls: wl.run(5, K)
    for wl in wls: wl.sync()
    best = 1e9
    for rep in range(int(os.environ.get('REPS', '5'))):
        be.device_sync()
        t0 = time.perf_counter()
        for _ in range(10):
            for wl in wls: wl.run(5, K)
        for wl in wls: wl.sync()
        dt = time.perf_counter() - t0
        best = min(best, dt)
    print(f"{name} S={S} B/S={B // S}: {best / (10 * K) * 1e6:8.1f} us/step  {B * 10 * K / best / 1e6:7.3f} M frames/s", flush=True)
    for wl in wls: wl.close()
