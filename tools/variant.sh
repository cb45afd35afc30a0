#!/bin/bash
# Build variant of the real-coefficient kernels next to the default library:  bash tools/variant.sh <name> "<-D flags>"
# -> htool_amd/libhmx_<name>.so (engine_f64 / engine_f32 recompiled with the flags, every other object shared with the default build;
# select it with HMX_LIB_PATH).  `make -C htool_amd/csrc` must have run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2
cd $ROOT/htool_amd/csrc
mkdir -p _obj_$name
HIPFLAGS="-std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags"
/opt/rocm/bin/hipcc $HIPFLAGS -DHMX_INST=0 -c engine_inst.hip -o _obj_$name/engine_f64.o &
/opt/rocm/bin/hipcc $HIPFLAGS -DHMX_INST=1 -c engine_inst.hip -o _obj_$name/engine_f32.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-soname,libhmx.so -o ../libhmx_$name.so _obj/cluster_tree.o _obj/block_tree.o _obj/geometry.o _obj/io.o \
  _obj/capi_host.o _obj/engine.o _obj_$name/engine_f64.o _obj_$name/engine_f32.o _obj/engine_z64.o _obj/engine_c32.o -lpthread
rm -rf _obj_$name
echo "htool_amd/libhmx_$name.so"
