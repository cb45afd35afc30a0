#!/bin/bash
# randomised parity sweeps with the alternative code paths forced through the environment (= the initial option values of every operator)
# usage: bash tools/fuzz_forced_paths.sh [seconds per sweep]
T=${1:-80}
run() { echo "== $*"; env "$@" python3 tools/fuzz_parity.py $T $SEED 2>&1 | tail -2; SEED=$((SEED+1)); }
SEED=41
run HMX_SYM_NO_VIEW=1
run HMX_SYM_MU_FUSED=1
run HMX_TRANS_STREAMS=0
run HMX_SYM_NO_VIEW=1 HMX_TRANS_STREAMS=0 FUZZ_RELEASE=1
run HMX_NO_MFMA=1 HMX_SYM_MU_FUSED=1
run HMX_SYM_EXPANDED=1 FUZZ_USER=1
run HMX_POOL_RANK_GUESS=1 FUZZ_ROUNDTRIP=1
run HMX_SR_MAX=128 HMX_R_TREE_PIECES=0 HMX_SORT_TASKS=0 HMX_REDUCE_WAVES=4 HMX_EXPAND_WAVES=8
run FUZZ_USER=1 FUZZ_RESERVE_GB=8
run HMX_SORT_TASKS=3 HMX_XCD_UNIT_ROWS=128 HMX_SYM_MU_FUSED=1
# round 6: groups of row ranges of the mirrored sweeps (sizes, no accumulators at all, a handful, the maximum), the expanded view (no longer the default)
run HMX_SYM_GROUP=1
run HMX_SYM_GROUP=2 HMX_SYM_GROUP_SLOTS=7
run HMX_SYM_GROUP=8 HMX_SYM_GROUP_SLOTS=1024 HMX_TRANS_STREAMS=0
run HMX_SYM_GROUP=16 HMX_SYM_GROUP_SLOTS=64 HMX_NO_MFMA=1
run HMX_SYM_GROUP=3 HMX_SYM_GROUP_SLOTS=0 HMX_SORT_TASKS=3
run HMX_SYM_MU_FUSED=0
run HMX_SYM_MU_FUSED=0 FUZZ_RESERVE_GB=8 FUZZ_USER=1
