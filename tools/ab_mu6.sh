#!/bin/bash
# same-box A/B of two builds of libhmx.so (HMX_LIB_PATH): bash tools/ab_mu6.sh <other.so>
OTHER=${1:-htool_amd/libhmx_prev.so}
run() { python3 bench.py $2 --steps 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for i in 1 2 3; do
  run new "--mu 16"
  HMX_LIB_PATH=$PWD/$OTHER run prev "--mu 16"
done
run new ""
HMX_LIB_PATH=$PWD/$OTHER run prev ""
