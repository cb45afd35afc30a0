#!/bin/bash
# quick check after a kernel change: the multi-RHS parity tests, then the probe on the reference operators
python -m pytest tests/test_gpu_parity.py -x -q -k "rhs or multi or sweeps or stored_triangle or transposed or coexist" 2>&1 | tail -3
python3 tools/probe.py --n 1000000 --sym S --mu 16 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0 2>&1 | grep "probe. {" | cut -c1-330
python3 tools/probe.py --n 1000000 --sym S --mu 16 --dtype f32 --variant sym_multi_rhs=1 --variant sym_multi_rhs=0 2>&1 | grep "probe. {" | cut -c1-330
python3 tools/probe.py --n 1000000 --mu 16 --trans T --variant transposed_layout=0 2>&1 | grep "probe. {" | cut -c1-330
python3 tools/probe.py --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --variant sym_multi_rhs=1 2>&1 | grep "probe. {" | cut -c1-330
