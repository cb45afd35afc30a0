#!/bin/bash
# R-stream chunk width histogram of a few configurations (HMX_BUILD_TIMING=1 prints it)
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --steps 3 --warmup 1 --no-cpu-baseline --mu 16"
for f in "$C5 --emulate-world 8 --emulate-rank 3" "$C5" "--steps 3 --warmup 1 --no-cpu-baseline" "--steps 3 --warmup 1 --no-cpu-baseline --emulate-world 8 --emulate-rank 3"; do
  echo "== $f"; HMX_BUILD_TIMING=1 python3 bench.py $f 2>&1 | grep "R-stream coefficients"
done
