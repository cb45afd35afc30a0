#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
F="--dtype z64 --steps 30 --no-cpu-baseline"
for rep in 1 2; do
  (cd $ROOT/ab_old/fbbeafb && python3 bench.py $F 2>/dev/null | show r2)
  (cd $ROOT && python3 bench.py $F 2>/dev/null | show HEAD)
  (cd $ROOT && HMX_R_TREE_PIECES=0 python3 bench.py $F 2>/dev/null | show HEAD_steps)
  (cd $ROOT && HMX_LAYOUT_THREADS=1 python3 bench.py $F 2>/dev/null | show HEAD_1thread)
done
