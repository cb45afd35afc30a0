cd /root/repo
for P in "--n 1000000 --mu 16" "--n 1000000" "--n 1000000 --sym S --mu 16 --variant sym_multi_rhs=1" "--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3 --variant default --variant sym_multi_rhs=1" "--n 1000000 --sym S"; do
 for O in "task_order=1" "task_order=3" "task_order=3 --build-option xcd_unit_rows=2048" "task_order=0"; do
  echo "== $P | $O"
  python3 tools/probe.py $P --build-option $O 2>&1 | grep "probe. {" | cut -c1-420
 done
done
