#!/bin/bash
# the previous round's final revision against the working tree, same box, the configurations DESIGN.md quotes
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for flags in "" "--sym S" "--dtype f32" "--dtype z64" "--trans T" "--mu 8" "--dtype c32"; do
  echo "== $flags"; bash $ROOT/tools/ab_builds.sh run "$flags"
done
