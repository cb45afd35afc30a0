#!/bin/bash
# Runs ON THE GPU BOX: rows per R-stream piece (r_piece_rows) 512 against 1024 / 2048 on the BASELINE workloads, twice each (written arrays placed: 170 GB slab)
cd /root/repo
for P in "--n 1000000" "--n 1000000 --sym S" "--n 1000000 --mu 16" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3"; do
 for O in 512 1024 2048 512 1024 2048; do
  echo "== $P | r_piece_rows=$O"
  python3 tools/probe.py $P --reserve-gb 170 --build-option r_piece_rows=$O 2>&1 | grep "probe. {" | cut -c1-330
 done
done
