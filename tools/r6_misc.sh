#!/bin/bash
set -u
mkdir -p gpurun_out/r6_misc
O=gpurun_out/r6_misc
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_c_abi.py -m gpu -x -q -k "output_vectors or placed_by or c_level_distributed or sweeps_of_32" > $O/tests.log 2>&1; tail -4 $O/tests.log
bash tools/rank_step_overhead.sh 2>&1 | tee $O/step_overhead.log | cut -c1-330
HMX_BUILD_TIMING=1 python3 bench.py --steps 5 --no-cpu-baseline --no-callback-build > $O/bench_bt.json 2> $O/bench_bt.err; grep "hmx build" $O/bench_bt.err | cut -c1-150 | head -60
