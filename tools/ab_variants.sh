#!/bin/bash
# every htool_amd/libhmx_*.so variant (tools/variant.sh) against the default build on one box: bash tools/ab_variants.sh "<bench flags>" [env=val ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
flags=$1; shift
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % '$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
echo "== $flags $*"
for lib in $ROOT/htool_amd/libhmx.so $ROOT/htool_amd/libhmx_*.so; do
  env HMX_LIB_PATH=$lib "$@" python3 $ROOT/bench.py $flags --no-cpu-baseline --no-callback-build 2>/dev/null | show $(basename $lib .so)
done
