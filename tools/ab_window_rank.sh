#!/bin/bash
# window reduce kernels (X rows in LDS) on ONE rank's share of a row-partitioned operator, where X traffic per stream byte is 8 x higher
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --steps 30 --no-cpu-baseline --mu 16 --emulate-world 8 --emulate-rank 3"
C4="--steps 30 --no-cpu-baseline --mu 16 --emulate-world 8 --emulate-rank 3"
for rep in 1 2; do
  python3 bench.py $C5 2>/dev/null | show "c5 staged"
  HMX_MU_WINDOW=1 python3 bench.py $C5 2>/dev/null | show "c5 window"
  python3 bench.py $C4 2>/dev/null | show "1e6/8 staged"
  HMX_MU_WINDOW=1 python3 bench.py $C4 2>/dev/null | show "1e6/8 window"
done
