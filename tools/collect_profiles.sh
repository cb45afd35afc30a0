#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 evidence of the bench command, written under gpurun_out/<tag>/.  Kernel-trace statistics and
# the PMC counters are collected in SEPARATE runs (counters never together with other trace domains).
#   usage: bash tools/collect_profiles.sh <tag> [bench.py flags]      e.g. r2_n1e6   |   r2_sym --sym S
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
FLAGS="--no-cpu-baseline $*"
python3 bench.py --steps 30 $FLAGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 $FLAGS > $OUT/under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 3 --warmup 1 $FLAGS > /dev/null 2> $OUT/pmc_$c.err
  python3 tools/pmc_summary.py $OUT/pmc_$c --json > $OUT/pmc_$c.json
done
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -c 600 $OUT/bench.json; echo; head -12 $OUT/kernel_stats.csv | cut -c1-160
