#!/bin/bash
set -u
mkdir -p gpurun_out/r6_second
O=gpurun_out/r6_second
python3 tools/output_place_probe.py > $O/output_place.log 2>&1
grep "output place" $O/output_place.log | cut -c1-420
python3 bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
head -c 1200 $O/bench.json; echo; tail -3 $O/bench.err
python3 bench.py --steps 20 --no-cpu-baseline --output-placement torch > $O/bench_torch.json 2> $O/bench_torch.err
head -c 400 $O/bench_torch.json; echo
timeout 900 python3 -m pytest tests -m gpu -x -q -k "placed or slab or parity or c_abi" > $O/tests.log 2>&1; tail -5 $O/tests.log
