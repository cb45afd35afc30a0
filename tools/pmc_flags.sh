#!/bin/bash
# Runs ON THE GPU BOX: matrix-core, LDS and wait counters of `bench.py <flags>` (separate rocprofv3 --pmc passes, kernel trace only).
#   bash tools/pmc_flags.sh <tag> <bench flags ...>
TAG=$1; shift
OUT=$PWD/gpurun_out/${TAG}_counters
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-callback-build "$@" > /dev/null 2> $OUT/$tag.err
  echo "== $set" >> $OUT/summary.txt
  python3 tools/pmc_summary.py $OUT/$tag >> $OUT/summary.txt 2>&1
  rm -rf $OUT/$tag
done
cat $OUT/summary.txt
