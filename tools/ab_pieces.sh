#!/bin/bash
# R-stream pieces along the source tree (default) or in fixed steps, same box: bash tools/ab_pieces.sh
run() { python3 bench.py $2 --steps 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "" "--sym S" "--mu 16" "--dtype f32"; do
  for rep in 1 2; do
    HMX_R_TREE_PIECES=1 run tree "$flags"
    HMX_R_TREE_PIECES=0 run steps "$flags"
  done
done
