#!/bin/bash
# timing of build variants (tools/variant.sh) on one box
for P in "--n 1000000 --mu 16 --dtype f32" "--n 1000000 --sym S --mu 16 --dtype f32 --variant sym_multi_rhs=1" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3 --variant default --variant sym_multi_rhs=1"; do
for lib in htool_amd/libhmx.so htool_amd/libhmx_*.so; do
  echo "== $lib $P"
  HMX_LIB_PATH=$PWD/$lib python3 tools/probe.py $P 2>&1 | grep "probe. {" | cut -c1-400
done
done
