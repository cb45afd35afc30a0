#!/bin/bash
# Same-box comparison of several REVISIONS of this repository (how the load-shape regression of DESIGN.md 4b was bisected).
#   here (no GPU):  bash tools/ab_builds.sh build <sha> [<sha> ...]     -> ab_old/<sha>/ = that revision's package + bench.py, libhmx.so built
#   on the GPU box: bash tools/ab_builds.sh run "<bench flags>"          -> one line per revision in ab_old/ and for the working tree, twice
# ab_old/ is not tracked (.git/info/exclude) but travels with the gpurun snapshot.
set -u
cmd=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$cmd" = build ]; then
  mkdir -p $ROOT/ab_old
  grep -qx "ab_old/" $ROOT/.git/info/exclude 2>/dev/null || echo "ab_old/" >> $ROOT/.git/info/exclude
  for sha in "$@"; do
    git -C $ROOT worktree add -f /tmp/w_$sha $sha -q || exit 1
    (cd /tmp/w_$sha/htool_amd/csrc && make -j3 > /dev/null 2>&1) || { echo "build of $sha failed"; exit 1; }
    rm -rf /tmp/w_$sha/htool_amd/csrc/_obj $ROOT/ab_old/$sha
    mkdir -p $ROOT/ab_old/$sha
    cp -r /tmp/w_$sha/htool_amd /tmp/w_$sha/bench.py /tmp/w_$sha/oracle /tmp/w_$sha/profiles $ROOT/ab_old/$sha/
    git -C $ROOT worktree remove --force /tmp/w_$sha
    echo "ab_old/$sha ready"
  done
else
  FLAGS="$* --steps 30 --no-cpu-baseline"
  show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
  for rep in 1 2; do
    for dir in $ROOT/ab_old/*/; do
      [ -f $dir/bench.py ] && (cd $dir && python3 bench.py $FLAGS 2>/dev/null | show $(basename $dir))
    done
    (cd $ROOT && python3 bench.py $FLAGS 2>/dev/null | show HEAD)
  done
fi
