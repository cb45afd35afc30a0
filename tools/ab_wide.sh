#!/bin/bash
# sweeps of 32 right-hand sides (HMX_MFMA_WIDE, default on) against sweeps of 16, same box
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "--mu 32" "--mu 17" "--mu 24" "--mu 64" "--dtype f32 --mu 32" "--sym S --mu 32"; do
  echo "== $flags"
  for rep in 1 2; do
    HMX_MFMA_WIDE=0 python3 bench.py $flags --steps 20 --no-cpu-baseline 2>/dev/null | show "16-wide"
    HMX_MFMA_WIDE=1 python3 bench.py $flags --steps 20 --no-cpu-baseline 2>/dev/null | show "32-wide"
  done
done
