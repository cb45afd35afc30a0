#!/bin/bash
# ACA kernel time of the working tree against the revisions in ab_old/ (tools/ab_builds.sh build <sha>), same box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
one() { (cd $1 && HMX_BUILD_TIMING=1 python3 bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "compression kernels\|round 0" | sed "s/^/$2 /"); }
for flags in "" "--n 4000000 --sym S --dtype f32 --eps 1e-6" "--dtype z64 --sym H"; do
  echo "== $flags"
  for rep in 1 2; do
    for dir in $ROOT/ab_old/*/; do one $dir $(basename $dir) "$flags"; done
    one $ROOT HEAD "$flags"
  done
done
