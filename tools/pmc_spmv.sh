#!/bin/bash
# HBM traffic of the SpMV kernel from hardware counters, one rocprofv3 --pmc pass per counter (MI355X guide:
# separate passes, no tracing flags next to --pmc).  Run on the GPU box from the repo root:
#   bash tools/pmc_spmv.sh [cells]      -> gpurun_out/pmc_spmv_n<cells>.json (+ raw counter averages)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
set -u
N=${1:-59}
OUT=gpurun_out/pmc_n$N
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
for C in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -- python3 tools/tune_spmv.py --cells "$N" --rounds 1 --reps 5 \
    --grids 0 --variants 3,13 --unrolls 5 > "$OUT/$C.log" 2>&1 || echo "pass $C failed (see $OUT/$C.log)"
done
python3 tools/pmc_reduce.py "$OUT" "$N" > "gpurun_out/pmc_spmv_n$N.json"
cat "gpurun_out/pmc_spmv_n$N.json"
find "$OUT" -name "*.csv" -size +5M -delete
