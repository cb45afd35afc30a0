#!/bin/bash
# config 5 on one GPU and on one rank's share, fused stored-triangle product against the expanded view, one box
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --no-cpu-baseline --no-callback-build --steps 10"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()}, 'streams GB', d['config'].get('algorithmic_GB'))"; }
HMX_SYM_MU_FUSED=1 python3 bench.py $C5 2>/dev/null | show "c5 fused   "
HMX_SYM_MU_FUSED=0 python3 bench.py $C5 2>/dev/null | show "c5 expanded"
python3 bench.py $C5 --emulate-world 8 --emulate-rank 3 2>/dev/null | show "c5 rank 3/8"
N6="--sym S --mu 16 --no-cpu-baseline --no-callback-build"
HMX_SYM_MU_FUSED=1 python3 bench.py $N6 2>/dev/null | show "1e6 S fused   "
HMX_SYM_MU_FUSED=0 python3 bench.py $N6 2>/dev/null | show "1e6 S expanded"
HMX_SYM_MU_FUSED=1 python3 bench.py $N6 --dtype f32 2>/dev/null | show "1e6 S f32 fused   "
HMX_SYM_MU_FUSED=0 python3 bench.py $N6 --dtype f32 2>/dev/null | show "1e6 S f32 expanded"
