#!/bin/bash
# first GPU call of round 6: baselines of the workloads the round works on (one build each, per-kernel times), matrix-core counters, the modes probe
set -u
mkdir -p gpurun_out/r6_first
O=gpurun_out/r6_first
python3 tools/probe.py --sym S --mu 16 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0 > $O/sym_mu16.log 2>&1
tail -4 $O/sym_mu16.log | cut -c1-900
python3 tools/probe.py --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3 --variant default --variant sym_multi_rhs=1 > $O/c5.log 2>&1
tail -4 $O/c5.log | cut -c1-900
python3 tools/probe.py --sym S --variant default > $O/sym.log 2>&1
tail -2 $O/sym.log | cut -c1-900
bash tools/collect_mfma.sh r6_mfma_mu16 F64 --mu 16
bash tools/collect_mfma.sh r6_mfma_c5_rank3 F32 --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
bash tools/modes_probe.sh
