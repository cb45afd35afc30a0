#!/bin/bash
# Runs ON THE GPU BOX: texture-addresser / L1 / L2 counters of `bench.py <flags>` (separate rocprofv3 --pmc passes, kernel trace only).
#   bash tools/pmc_mem.sh <tag> <bench flags ...>
TAG=$1; shift
OUT=$PWD/gpurun_out/${TAG}_memcounters
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TCC|TD|SQ|SPI|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/avail.txt
: > $OUT/summary.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-callback-build "$@" > /dev/null 2> $OUT/$tag.err
  echo "== $set" >> $OUT/summary.txt
  python3 tools/pmc_summary.py $OUT/$tag 2>&1 | grep -A2 "expand\|reduce_\|rowsym" >> $OUT/summary.txt
  tail -2 $OUT/$tag.err | grep -i "error\|invalid\|not" >> $OUT/summary.txt
  rm -rf $OUT/$tag
done
cat $OUT/summary.txt; wc -l $OUT/avail.txt
