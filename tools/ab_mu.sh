#!/bin/bash
# A/B of the multi-RHS kernel variants (window reduce stage, grouped expand stage) on one box: bench lines into gpurun_out/
out=gpurun_out/${1:-r3_ab_mu}.log
: > $out
run() { # name, env assignments..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "== $name :: $*" >> $out
  env "${envs[@]}" python bench.py --steps 20 --no-cpu-baseline --no-reference "$@" 2>>${out%.log}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" >> $out
}
for args in "--mu 16" "--dtype z64 --mu 8"; do
  run default -- $args
  run nowindow HMX_MU_WINDOW=0 -- $args
  run nogroups HMX_MU_GROUPS=0 -- $args
  run neither HMX_MU_WINDOW=0 HMX_MU_GROUPS=0 HMX_E_GROUPS=0 -- $args
done
run default -- 
run f32_default -- --dtype f32 --mu 16
cat $out
