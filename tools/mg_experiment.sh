# usage: bash tools/mg_experiment.sh  -- block-Jacobi Chebyshev smoother: interval and degree
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
for n in 34 59; do for cfg in "0 20 2" "1 20 2" "1 10 2" "1 15 2" "1 30 2" "1 20 3" "1 30 3"; do set -- $cfg; echo "cells=$n block=$1 ratio=$2 nu=$3"; MI_MG_BLOCK=$1 MI_MG_RATIO=$2 MI_MG_NU=$3 MI_MG_NU_COARSE=$3 python bench.py --cells $n --steps 3 --warmup 1 --cpu-cells 0 2>/dev/null | python tools/summarize_bench.py | cut -c1-110; done; done
