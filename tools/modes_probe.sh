#!/bin/bash
# Runs ON THE GPU BOX: the two speeds of the expand kernels.  The same operator built several times in one process
# (tools/placement_builds.py: every build's streams lie elsewhere), once plain (HIP-event times per build) and once under
# rocprofv3 --pmc with the L2 <-> fabric read requests PER CHANNEL (json output keeps the counter's instances apart).
set -u
OUT=$PWD/gpurun_out/r6_modes
mkdir -p $OUT
export TMPDIR=/tmp
HMX_BUILD_TIMING=1 python3 tools/placement_builds.py > $OUT/plain.log 2>&1
grep -E "placement|arrays:" $OUT/plain.log | cut -c1-400 | tail -24
for c in TCC_EA0_RDREQ TCC_REQ; do
  HMX_BUILD_TIMING=1 HMX_PLACEMENT_BUILDS=5 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "expand_mfma16s|expand_kernel" --output-format json csv -d $OUT/pmc_$c \
      -- python3 tools/placement_builds.py > $OUT/pmc_$c.log 2>&1
  python3 tools/modes_summary.py $OUT/pmc_$c $c > $OUT/modes_$c.json 2> $OUT/modes_$c.err
  find $OUT/pmc_$c -name "*.json" -size +40M -delete
  head -c 1500 $OUT/modes_$c.json; echo
done
