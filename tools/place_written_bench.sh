cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -x -k "written_arrays or coexist or matvec_matches" 2>&1 | tail -4
for F in "" "--sym S" "--mu 16" "--sym S --mu 16 --option sym_multi_rhs=1" "--sym S --mu 16 --option sym_multi_rhs=0" "--trans T"; do
  for O in 1 0; do
  echo "== bench $F place_written=$O"
  python3 bench.py --steps 20 --no-cpu-baseline $F --option place_written=$O 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/8000,3), {k:round(v,3) for k,v in d['roofline']['kernels_ms'].items()}, d['compress']['written_array_placement'], 'reserve_s', round(d['compress']['reserve_s'],2))"
  done
done
