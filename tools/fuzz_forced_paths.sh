#!/bin/bash
# randomised parity sweeps with the alternative code paths forced through the environment (initial option values of every operator)
set -x
HMX_SYM_NO_VIEW=1 python3 tools/fuzz_parity.py 100 31 2>&1 | tail -3
HMX_SYM_MU_FUSED=1 python3 tools/fuzz_parity.py 80 32 2>&1 | tail -3
HMX_TRANS_STREAMS=0 python3 tools/fuzz_parity.py 80 33 2>&1 | tail -3
HMX_SYM_NO_VIEW=1 HMX_TRANS_STREAMS=0 FUZZ_RELEASE=1 python3 tools/fuzz_parity.py 80 34 2>&1 | tail -3
python3 tools/fuzz_parity.py 80 35 2>&1 | tail -3
