#!/bin/bash
# Runs ON THE GPU BOX (gpurun): matrix-core utilisation of the multi-RHS kernels (north_star: "MFMA utilisation against peak").
# One rocprofv3 pass per workload with SQ_VALU_MFMA_BUSY_CYCLES, the MFMA op counter of the coefficient type and GRBM_GUI_ACTIVE
# (counters only next to --kernel-trace; never with other trace domains), summarised by tools/mfma_summary.py.
#   usage: bash tools/collect_mfma.sh <tag> <F64|F32> [bench.py flags]
set -u
TAG=$1; TY=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_$TY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc \
    -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline $* > $OUT/under_pmc.json 2> $OUT/pmc.err
python3 tools/mfma_summary.py $OUT/pmc $TY > $OUT/mfma_pmc_summary.json
rm -rf $OUT/pmc
cat $OUT/mfma_pmc_summary.json | head -60
