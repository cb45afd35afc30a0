#!/bin/bash
# config 5 (N=4e6 fp32 'S' eps=1e-6): kernel times of the whole operator and of one rank's share, mu = 16 and mu = 1
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],3), 'ms', round(d['config']['algorithmic_GB'],2), 'GB alg', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --steps 20 --no-cpu-baseline"
python3 bench.py $C5 --mu 16 --emulate-world 8 --emulate-rank 3 2>/dev/null | show rank3_mu16
python3 bench.py $C5 --mu 1 --emulate-world 8 --emulate-rank 3 2>/dev/null | show rank3_mu1
python3 bench.py $C5 --mu 16 2>/dev/null | show whole_mu16
python3 bench.py $C5 --mu 1 2>/dev/null | show whole_mu1
