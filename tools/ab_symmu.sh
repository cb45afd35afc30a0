#!/bin/bash
# stored-triangle multi-RHS kernels: build variants (htool_amd/libhmx_<name>.so, HIPFLAGS_EXTRA=-DHMX_SYMMU_PT=.. / -DHMX_SYMMU_SINGLE=1 /
# -DHMX_ROWSYM_WAVES=..) against the default build, one box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "--sym S --mu 16" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --steps 10"; do
  echo "== $flags"
  for lib in libhmx libhmx_pt18 libhmx_pt20 libhmx_single libhmx_single_pt20 libhmx_rw2 libhmx_rw8; do
    [ -f $ROOT/htool_amd/$lib.so ] || continue
    HMX_LIB_PATH=$ROOT/htool_amd/$lib.so HMX_SYM_MU_FUSED=1 python3 bench.py $flags --no-cpu-baseline --no-callback-build 2>/dev/null | show $lib
  done
done
