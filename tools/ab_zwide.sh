#!/bin/bash
# complex right-hand sides: expand stage in sweeps of 16 (default) against sweeps of 8, same box
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "--dtype z64 --mu 16" "--dtype z64 --mu 11" "--dtype c32 --mu 16"; do
  echo "== $flags"
  for rep in 1 2; do
    HMX_MFMA_WIDE=0 python3 bench.py $flags --steps 20 --no-cpu-baseline 2>/dev/null | show "8-wide"
    HMX_MFMA_WIDE=1 python3 bench.py $flags --steps 20 --no-cpu-baseline 2>/dev/null | show "16-wide expand"
  done
done
