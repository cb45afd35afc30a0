#!/bin/bash
# A/B of the fp64 mu = 16 kernel variants on one box (bench lines into gpurun_out/)
out=gpurun_out/${1:-r3_ab_stage}.log
: > $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name :: $*" >> $out
  env "${envs[@]}" python bench.py --steps 20 --no-cpu-baseline --no-reference "$@" 2>>${out%.log}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" >> $out
}
run staged -- --mu 16
run staged_window HMX_MU_WINDOW=1 -- --mu 16
run direct HMX_MFMA_STAGE=0 HMX_MU_GROUPS=0 HMX_MU_WINDOW=0 -- --mu 16
run staged_f32 HMX_MFMA_F32=1 -- --mu 16 --dtype f32
run valu_f32 -- --mu 16 --dtype f32
run staged_sym -- --mu 16 --sym S
run staged_n1e5 -- --mu 16 --n 100000
run staged_mu32 -- --mu 32
cat $out
