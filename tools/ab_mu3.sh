#!/bin/bash
out=gpurun_out/${1:-r3_ab_valu}.log
: > $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name :: $*" >> $out
  env "${envs[@]}" python bench.py --steps 20 --no-cpu-baseline --no-reference "$@" 2>>${out%.log}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" >> $out
}
run f32 -- --mu 16 --dtype f32
run z64 -- --mu 8 --dtype z64
run c32 -- --mu 8 --dtype c32
run f64mu8 -- --mu 8
run cfg5 -- --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16
cat $out
