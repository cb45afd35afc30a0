#!/bin/bash
# round 5 baseline on one box: configs[4] rank share (mu = 16 fp32), kernel times, layout statistics, VALU comparison
show() { python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', round(d['ms_per_step'],3), 'ms', round(d['config']['algorithmic_GB'],2), 'GB alg', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()}, 'frac', round(d['config']['hbm_roofline_frac'],3))"; }
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --steps 20 --no-cpu-baseline --no-callback-build"
HMX_BUILD_TIMING=1 python3 bench.py $C5 --mu 16 --emulate-world 8 --emulate-rank 3 2>gpurun_out/r5_base_rank3.err | show rank3_mu16
grep -i "R-stream coeff\|layout:" gpurun_out/r5_base_rank3.err | head -40
HMX_NO_MFMA=1 python3 bench.py $C5 --mu 16 --emulate-world 8 --emulate-rank 3 2>/dev/null | show rank3_mu16_valu
python3 bench.py $C5 --mu 1 --emulate-world 8 --emulate-rank 3 2>/dev/null | show rank3_mu1
python3 bench.py $C5 --mu 16 --emulate-world 8 --emulate-rank 0 2>/dev/null | show rank0_mu16
N6="--sym S --mu 16 --no-cpu-baseline --no-callback-build"
HMX_SYM_MU_FUSED=1 python3 bench.py $N6 2>/dev/null | show "1e6_S_mu16_fused"
HMX_SYM_MU_FUSED=0 python3 bench.py $N6 2>/dev/null | show "1e6_S_mu16_view"
python3 bench.py --sym S --no-cpu-baseline --no-callback-build 2>/dev/null | show "1e6_S_mu1"
python3 bench.py --mu 16 --no-cpu-baseline --no-callback-build 2>/dev/null | show "1e6_N_mu16"
python3 bench.py --mu 16 --dtype f32 --no-cpu-baseline --no-callback-build 2>/dev/null | show "1e6_N_mu16_f32"
