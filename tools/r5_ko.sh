#!/bin/bash
# timing of build variants (tools/variant.sh) on one box
for P in "--n 1000000 --sym S --mu 16 --variant sym_multi_rhs=1" "--n 1000000 --mu 16" ; do
for lib in htool_amd/libhmx.so htool_amd/libhmx_*.so; do
  echo "== $lib $P"
  HMX_LIB_PATH=$PWD/$lib python3 tools/probe.py $P 2>&1 | grep "probe. {" | cut -c1-330
done
done
