#!/bin/bash
# round-5 evidence on one box: bench lines + rocprofv3 kernel-trace statistics + PMC traffic for the headline and the other BASELINE workloads
bash tools/collect_profiles.sh r5_n1e6
bash tools/collect_profiles.sh r5_sym --sym S
bash tools/collect_profiles.sh r5_mu16 --mu 16
bash tools/collect_profiles.sh r5_transT --trans T
bash tools/collect_profiles.sh r5_sym_mu16_stored --sym S --mu 16 --option sym_multi_rhs=1
bash tools/collect_profiles.sh r5_sym_mu16_view --sym S --mu 16 --option sym_multi_rhs=0
bash tools/collect_profiles.sh r5_c5_rank3 --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
