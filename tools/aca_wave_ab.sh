#!/bin/bash
# Runs ON THE GPU BOX: the compression of the headline operator with and without the one-wave kernel for small blocks (aca_wave_max), phase times on stderr
for O in 0 256; do
  echo "== aca_wave_max=$O"
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --option aca_wave_max=$O --option build_timing=1 2>&1 | grep -E "hmx build.*(round|compression|pool|ACA|aca)" | cut -c1-220
done
