export TMPDIR=/tmp
mkdir -p gpurun_out/r2p
for s in 1 0; do
  HMX_SORT_TASKS=$s rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/pmc_s$s -- python3 bench.py --mu 16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2p/err$s
  python3 tools/pmc_summary.py gpurun_out/r2p/pmc_s$s --json > gpurun_out/r2p/pmc_s$s.json
  rm -rf gpurun_out/r2p/pmc_s$s
  python3 -c "
import json; f=json.load(open('gpurun_out/r2p/pmc_s$s.json'))
for k in f:
    if 'mfma16' in k: print('sort $s', k[:50], 'fetch GB', round(2*1024*f[k]['FETCH_SIZE']['mean']/1e9,2))"
done
