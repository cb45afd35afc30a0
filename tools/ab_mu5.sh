#!/bin/bash
out=gpurun_out/${1:-r3_ab_zmfma}.log
: > $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name :: $*" >> $out
  env "${envs[@]}" python bench.py --steps 20 --no-cpu-baseline --no-reference "$@" 2>>${out%.log}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" >> $out
}
run z64_mfma -- --dtype z64 --mu 8
run z64_valu HMX_NO_MFMA=1 -- --dtype z64 --mu 8
run c32_mfma -- --dtype c32 --mu 8
run c32_valu HMX_NO_MFMA=1 -- --dtype c32 --mu 8
run z64_mu1 -- --dtype z64
cat $out
