#!/bin/bash
# Runs ON THE GPU BOX: matrix-core, LDS and wait counters of the fp64 --mu <n> product (argument, default 16) (separate rocprofv3 --pmc passes, kernel trace only)
MU=${1:-16}
OUT=$PWD/gpurun_out/r3_mu${MU}_counters
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --mu $MU --no-cpu-baseline --no-reference > /dev/null 2> $OUT/$tag.err
  echo "== $set" >> $OUT/summary.txt
  python3 tools/pmc_summary.py $OUT/$tag >> $OUT/summary.txt 2>&1
  rm -rf $OUT/$tag
done
cat $OUT/summary.txt
