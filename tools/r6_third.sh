#!/bin/bash
set -u
mkdir -p gpurun_out/r6_third
O=gpurun_out/r6_third
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_complex.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests.log 2>&1; tail -8 $O/tests.log
python3 tools/probe.py --sym S --mu 16 --variant sym_multi_rhs=1 > $O/sym_mu16.log 2>&1
tail -1 $O/sym_mu16.log | cut -c1-600
python3 tools/probe.py --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3 --variant sym_multi_rhs=1 > $O/c5.log 2>&1
tail -1 $O/c5.log | cut -c1-600
python3 tools/probe.py --mu 16 --trans T --variant transposed_layout=0 > $O/transT_mu16.log 2>&1
tail -1 $O/transT_mu16.log | cut -c1-600
