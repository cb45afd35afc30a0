#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one kernel for two builds of the library (A/B of a code change that is invisible in the ISA statistics)
#   usage: bash tools/r6_pmc_ab.sh <kernel regex> <lib A> <lib B> [probe flags]
set -u
K=$1; A=$2; B=$3; shift 3
export TMPDIR=/tmp
for lib in $A $B; do
  O=$PWD/gpurun_out/r6_pmc_ab/$(basename $lib .so)
  mkdir -p $O
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $set | cut -d' ' -f1)
    HMX_LIB_PATH=$PWD/htool_amd/$lib rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$K" --output-format csv -d $O/$tag -- python3 tools/probe.py "$@" > $O/$tag.log 2>&1
    python3 tools/pmc_summary.py $O/$tag | grep -A 12 "$K" | head -14
    rm -rf $O/$tag
  done
done
