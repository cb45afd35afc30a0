#!/usr/bin/env python3
"""Tuning sweep over launch-geometry knobs of the matvec kernels (env vars read by libhmx at first use).
Usage on the GPU box: python tools/sweep.py  (spawns one bench.py per configuration)."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = [dict(HMX_REDUCE_WAVES=r, HMX_EXPAND_WAVES=e, HMX_SORT_TASKS=s) for r, e, s in
           [(4, 4, 1)] * 5]
extra = sys.argv[1:]
for c in configs:
    env = dict(os.environ, **{k: str(v) for k, v in c.items()})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = d["roofline"]["kernels_ms"]
        print(c, "value %.0f GB/s  step %.3f ms  reduce %.3f combine %.3f expand %.3f" % (d["value"], d["ms_per_step"], k.get("reduce_kernel", 0), k.get("combine_kernel", 0), k.get("expand_kernel", 0)), flush=True)
    except Exception as ex:
        print(c, "FAILED", ex, out.stderr[-500:], flush=True)
