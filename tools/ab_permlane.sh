#!/bin/bash
# expand stage of the multi-RHS product: operand layout by lane swaps (default) against the LDS-staged form (libhmx_staged.so,
# -DHMX_EXPAND_PERMLANE=0), one box, alternating
ROOT=$(cd "$(dirname "$0")/.." && pwd)
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for flags in "--mu 16" "--mu 16 --dtype f32" "--mu 12" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3" "--mu 16 --emulate-world 8 --emulate-rank 3"; do
  echo "== $flags"
  for rep in 1 2; do
    for lib in libhmx libhmx_staged; do
      HMX_LIB_PATH=$ROOT/htool_amd/$lib.so python3 bench.py $flags --steps 30 --no-cpu-baseline --no-callback-build 2>/dev/null | show $lib
    done
  done
done
