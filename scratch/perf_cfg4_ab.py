import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for n in sys.argv[1:]:
        env = dict(os.environ, DSENH_LIB=os.path.join(ROOT, "scratch", "variants", "libdsenh_%s.so" % n))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_cfg4.py")], env=env, capture_output=True, text=True)
        vals = [json.loads(l)["value"] for l in r.stdout.strip().splitlines() if l.startswith("{")]
        print("%-8s cfg4 T=1 %.3f M  T=39 %.3f M" % (n, vals[0] / 1e6, vals[1] / 1e6) if len(vals) == 2 else (n, r.stderr[-400:]), flush=True)
