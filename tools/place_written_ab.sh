#!/bin/bash
# Runs ON THE GPU BOX: the written arrays of the products where first fit puts them (place_written=0) against where the placement probe puts
# them (1), for two sizes of the reserved slab
cd /root/repo
for P in "--n 1000000" "--n 1000000 --sym S" "--n 1000000 --mu 16" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3"; do
 for R in 64 170; do
 for O in 0 1; do
  echo "== $P | reserve $R GB | place_written=$O"
  python3 tools/probe.py $P --reserve-gb $R --build-option place_written=$O --build-option build_timing=1 2>&1 | grep -E "probe. \{|arrays:" | cut -c1-400
 done
 done
done
