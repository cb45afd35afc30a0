#!/bin/bash
# Runs ON THE GPU BOX: rows per R-stream piece (r_piece_rows) 512 against 1024 on the BASELINE workloads and the symmetric / transposed forms, twice each
cd /root/repo
for P in "--n 1000000" "--n 1000000 --sym S" "--n 1000000 --trans T" "--n 1000000 --mu 16" "--n 1000000 --sym S --mu 16 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3"; do
 for O in 512 1024 512 1024; do
  echo "== $P | r_piece_rows=$O"
  python3 tools/probe.py $P --build-option r_piece_rows=$O 2>&1 | grep "probe. {" | cut -c1-330
 done
done
