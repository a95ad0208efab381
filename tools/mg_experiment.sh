# usage: bash tools/mg_experiment.sh  -- multigrid coarsening factor on the headline workload and on 20^3 / 34^3 blocks
for n in 59 34 20; do for cfg in "2 2" "3 2" "4 2" "3 3" "4 4"; do set -- $cfg; echo "cells=$n factor=$1 nu_coarse=$2"; MI_MG_FACTOR=$1 MI_MG_NU_COARSE=$2 python bench.py --cells $n --steps 3 --warmup 1 --cpu-cells 0 2>/dev/null | python tools/summarize_bench.py; done; done
