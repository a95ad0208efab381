# usage: bash tools/mg_experiment.sh  -- multigrid coarsest-level settings on the headline workload
for cfg in "1 12 60" "2 12 60" "4 12 60" "4 20 150" "8 16 100" "8 30 400"; do set -- $cfg; echo "coarsest=$1 degree=$2 ratio=$3"; MI_MG_COARSEST=$1 MI_MG_COARSE_DEGREE=$2 MI_MG_COARSE_RATIO=$3 python bench.py --steps 3 --warmup 1 --cpu-cells 0 2>/dev/null | python tools/summarize_bench.py; done
