import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line); print(round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["roofline"]["kernels_ms"].items()})
