#!/bin/bash
# Runs ON THE GPU BOX (one GPU): what a step costs beyond its kernels -- the launch + collective floor a SCALE record has to be read against.
#  (1) one rank's operator of BASELINE configs[3] (rank 3 of 8, N = 1e6) alone: step time against the sum of its kernels, eager launches;
#  (2) the whole N = 1e6 operator through hmx_dist_matvec_global_to_global with the collectives FORCED over a real one-rank RCCL communicator:
#      native C path (eager), torch.distributed path with the local kernels replayed from a HIP graph, and the same without the graph.
# The difference step - kernels is per step and does not depend on the operator's size.
set -u
O=$PWD/gpurun_out/r6_step_overhead
mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms']; s=sum(k.values())
print('%-44s step %.4f ms   kernels %.4f ms (%s)   step - kernels %.1f us   %s' % ('$1', d['ms_per_step'], s, ', '.join('%s %.3f' % (a.replace('_kernel',''), b) for a,b in k.items()), 1e3*(d['ms_per_step']-s), d['config'].get('parallelism','')))"; }
python3 bench.py --steps 200 --warmup 20 --emulate-world 8 --emulate-rank 3 --no-cpu-baseline --no-callback-build 2> $O/rank3.err | tee $O/rank3.json | show "rank 3 of 8 alone, eager"
python3 bench.py --steps 100 --warmup 10 --force-dist --no-cpu-baseline --no-callback-build 2> $O/force_native.err | tee $O/force_native.json | show "whole operator, forced RCCL (1 rank), native"
python3 bench.py --steps 100 --warmup 10 --force-dist --dist-impl python --no-cpu-baseline --no-callback-build 2> $O/force_graph.err | tee $O/force_graph.json | show "... torch.distributed + HIP graph of the kernels"
HMX_BENCH_NO_GRAPH=1 python3 bench.py --steps 100 --warmup 10 --force-dist --dist-impl python --no-cpu-baseline --no-callback-build 2> $O/force_eager.err | tee $O/force_eager.json | show "... torch.distributed, eager"
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-callback-build 2> $O/plain.err | tee $O/plain.json | show "whole operator, no collective"
