#!/bin/bash
# waves per workgroup of the staged matrix-core kernels, same box
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "--mu 16" "--dtype f32 --mu 16" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3"; do
  echo "== $flags"
  for rep in 1 2; do
    python3 bench.py $flags --steps 30 --no-cpu-baseline 2>/dev/null | show "x4 r4"
    HMX_MFMA_EXPAND_WAVES=2 python3 bench.py $flags --steps 30 --no-cpu-baseline 2>/dev/null | show "x2 r4"
    HMX_MFMA_EXPAND_WAVES=8 python3 bench.py $flags --steps 30 --no-cpu-baseline 2>/dev/null | show "x8 r4"
    HMX_MFMA_REDUCE_WAVES=1 python3 bench.py $flags --steps 30 --no-cpu-baseline 2>/dev/null | show "x4 r1"
    HMX_MFMA_REDUCE_WAVES=2 python3 bench.py $flags --steps 30 --no-cpu-baseline 2>/dev/null | show "x4 r2"
  done
done
