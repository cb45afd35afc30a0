#!/bin/bash
# round-6 evidence on one box: bench lines + rocprofv3 kernel-trace statistics + PMC traffic for the headline and the other BASELINE workloads
# (pass tags to collect only some: bash tools/collect_round.sh sym_mu16_stored c5_rank3)
R=r6
want() { [ $# -eq 0 ] && return 0; for t in "${SEL[@]}"; do [ "$t" = "$1" ] && return 0; done; return 1; }
SEL=("$@")
run() { tag=$1; shift; if [ ${#SEL[@]} -eq 0 ] || want $tag; then bash tools/collect_profiles.sh ${R}_$tag "$@"; fi; }
run n1e6
run sym --sym S
run mu16 --mu 16
run transT --trans T
run sym_mu16_stored --sym S --mu 16
run sym_mu16_view --sym S --mu 16 --option sym_multi_rhs=0
run c5_rank3 --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3
