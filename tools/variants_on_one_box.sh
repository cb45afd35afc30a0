#!/bin/bash
# every htool_amd/libhmx_*.so build variant (tools/variant.sh) against the default build on ONE box, through tools/probe.py
#   usage: bash tools/variants_on_one_box.sh "<probe flags>" ["<probe flags>" ...]
for P in "$@"; do
for lib in htool_amd/libhmx.so htool_amd/libhmx_*.so; do
  echo "== $lib $P"
  HMX_LIB_PATH=$PWD/$lib python3 tools/probe.py $P 2>&1 | grep "probe. {" | cut -c1-400
done
done
