# bench builds back to back (each process follows one that just released tens of GB): build time with and without the reserved slab
for r in default 0 default 0; do
  if [ $r = 0 ]; then export HMX_BENCH_RESERVE_GB=0; else unset HMX_BENCH_RESERVE_GB; fi
  python bench.py --no-cpu-baseline --no-reference --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1e6', d['ms_per_step'], d['compress'])"
done
for r in default 0; do
  if [ $r = 0 ]; then export HMX_BENCH_RESERVE_GB=0; else unset HMX_BENCH_RESERVE_GB; fi
  python bench.py --geom ball --no-cpu-baseline --no-reference --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ball', d['ms_per_step'], d['compress'])"
  python bench.py --dtype z64 --sym H --no-cpu-baseline --no-reference --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('herm', d['ms_per_step'], d['compress'])"
done
