#!/bin/bash
out=gpurun_out/${1:-r3_ab_cfg5}.log
: > $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name :: $*" >> $out
  env "${envs[@]}" python bench.py --steps 10 --no-cpu-baseline --no-reference "$@" 2>>${out%.log}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" >> $out
}
C5="--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16"
run valu -- $C5
run mfma_f32 HMX_MFMA_F32=1 -- $C5
run mfma_f32_rank HMX_MFMA_F32=1 -- --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
run valu_rank -- --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
cat $out
