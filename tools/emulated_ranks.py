#!/usr/bin/env python3
"""Runs ON THE GPU BOX: every rank's share of BASELINE configs[3] (N = 1e6 fp64, 8-way row partition, single vector) and configs[4]
(N = 4e6 fp32, 'S','L', sympartialACA eps = 1e-6, 16 right-hand sides, 8-way row partition) measured ALONE on the box's one GPU
(`bench.py --emulate-world 8 --emulate-rank k`: builds and times exactly what rank k of the 8-GPU run holds, no exchange), next to the
whole operator on one GPU.  What it yields is a PREDICTION of the local part of an 8-GPU step -- the slowest rank bounds it -- and of the
parallel efficiency before the output exchange; the pool has no 8-GPU node, so this is the only scaling evidence it can produce.
    python3 tools/emulated_ranks.py > gpurun_out/r5_emulated_ranks.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "configs[3]: N=1e6 fp64, partialACA eps=1e-4, single vector": ["--n", "1000000"],
    "configs[4]: N=4e6 fp32 S/L sympartialACA eps=1e-6, 16 right-hand sides": ["--n", "4000000", "--sym", "S", "--dtype", "f32", "--eps", "1e-6", "--mu", "16"],
}


def run(flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-callback-build", "--steps", "20", "--warmup", "5"] + flags,
                         capture_output=True, text=True, timeout=1200)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    return dict(ms=d["ms_per_step"], algorithmic_GB=d["config"]["algorithmic_GB"], GBps=d["value"], kernels_ms=d["roofline"]["kernels_ms"],
                stream_GB=None, build_s=d["compress"]["device_total_s"])


def main():
    res = {}
    for name, flags in CONFIGS.items():
        whole = run(flags)
        ranks = [run(flags + ["--emulate-world", "8", "--emulate-rank", str(k)]) for k in range(8)]
        slow = max(r["ms"] for r in ranks)
        res[name] = dict(one_gpu=whole, ranks=ranks, slowest_rank_ms=slow, sum_of_rank_GB=sum(r["algorithmic_GB"] for r in ranks),
                         predicted_local_speedup_8gpu=whole["ms"] / slow, predicted_parallel_efficiency_before_exchange=whole["ms"] / (8 * slow),
                         aggregate_GBps_before_exchange=sum(r["algorithmic_GB"] for r in ranks) / (slow * 1e-3),
                         note="prediction from single-GPU runs of each rank's operator; the exchange (all-gather of the output slices over xGMI) is not in it")
        print("[emulated ranks] %s: one GPU %.3f ms, ranks %s ms, slowest %.3f -> predicted efficiency %.2f" % (name, whole["ms"], [round(r["ms"], 3) for r in ranks], slow, whole["ms"] / (8 * slow)), file=sys.stderr, flush=True)
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
