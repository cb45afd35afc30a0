#!/bin/bash
# Runs ON THE GPU BOX: complex coefficients, several right-hand sides -- stored triangle / stored data against the expanded / transposed
# second layout, matrix-core kernels against the VALU kernels (profiles/r5_complex_stored.log)
P() { echo "== $*"; python3 tools/probe.py "$@" --check 2>&1 | grep "probe. {" | cut -c1-420; }
P --n 1000000 --sym H --dtype z64 --mu 8 --variant sym_multi_rhs=1 --variant sym_multi_rhs=1,matrix_cores=0 --variant sym_multi_rhs=0
P --n 1000000 --sym S --dtype z64 --mu 8 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0
P --n 1000000 --sym H --dtype c32 --mu 8 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0
P --n 1000000 --dtype z64 --mu 8 --trans C --variant transposed_layout=0 --variant transposed_layout=-1
