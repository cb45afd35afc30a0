#!/bin/bash
# host-generator build (bench.py --generator callback) at N = 1e6 for several thread / driver counts, one box: build phases on stderr
for cfg in "0 8" "0 4" "0 16" "8 8" "32 8"; do
  set -- $cfg
  echo "== threads $1 (0 = host cores) drivers $2"
  HMX_CALLBACK_DRIVERS=$2 HMX_BUILD_TIMING=1 python3 bench.py --no-cpu-baseline --generator callback --callback-threads $1 --steps 3 2>&1 >/dev/null | grep -E "round 0|compression kernels|dense blocks|device build"
done
