# A/B of the ACA team kernels on the high-rank Hermitian case (N=1e6 z64): thresholds, entries of a line per workgroup (0 = by the size of the launch)
for cfg in "4096 48 0" "2048 48 0" "2048 32 0" "4096 32 0" "8192 48 0"; do
  set -- $cfg
  echo "== TEAM_MIN=$1 TEAM_Q=$2 SLICE=$3"
  HMX_ACA_TEAM_MIN=$1 HMX_ACA_TEAM_Q=$2 HMX_ACA_TEAM_SLICE=$3 HMX_BUILD_TIMING=1 python bench.py --dtype z64 --sym H --no-cpu-baseline --no-reference --steps 3 2>&1 >/dev/null | grep -E "round|compression kernels|bench\]" | cut -c1-190
done
