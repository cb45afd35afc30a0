#!/bin/bash
# multi-RHS products on ONE rank's share and on the whole operator: working tree against ab_old/* (tools/ab_builds.sh)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for flags in "--mu 16 --emulate-world 8 --emulate-rank 3" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3" "--mu 16" "--dtype f32 --mu 16" "--mu 16 --emulate-world 2 --emulate-rank 1"; do
  echo "== $flags"; bash $ROOT/tools/ab_builds.sh run "$flags"
done
