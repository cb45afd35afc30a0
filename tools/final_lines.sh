#!/bin/bash
# Runs ON THE GPU BOX, after profiles/traffic.json has been refreshed for the committed sources: the bench lines once more, so that roofline.traffic
# (null while traffic.json's source hash lagged behind, as in the lines tools/collect_round.sh keeps) is filled.  First line: the driver's own command.
set -u
OUT=$PWD/gpurun_out/final_lines.jsonl
: > $OUT
line() { python3 bench.py "$@" 2>> $PWD/gpurun_out/final_lines.err | tail -1 >> $OUT; }
line
line --no-cpu-baseline --steps 30 --sym S
line --no-cpu-baseline --steps 30 --mu 16
line --no-cpu-baseline --steps 30 --trans T
line --no-cpu-baseline --steps 30 --sym S --mu 16
line --no-cpu-baseline --steps 30 --sym S --mu 16 --option sym_multi_rhs=0
line --no-cpu-baseline --steps 30 --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
cat $OUT
