#!/bin/bash
# HBM traffic of the bench's own kernels from hardware counters: one `rocprofv3 --pmc` pass per counter over the SAME
# command the bench line comes from (MI355X guide: separate passes, no tracing flags next to --pmc).
#   bash tools/pmc_bench.sh [outdir]   -> <outdir>/pmc_bench.json
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
set -u
FAILED=0
OUT=${1:-gpurun_out/pmc_bench}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
for C in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -- python3 bench.py --steps 1 --warmup 0 --cpu-cells 0 \
    > "$OUT/$C.log" 2>&1 || { echo "pass $C failed (see $OUT/$C.log)"; FAILED=1; }
done
python3 tools/pmc_bench_reduce.py "$OUT" > "$OUT/pmc_bench.json" || FAILED=1
[ -s "$OUT/pmc_bench.json" ] || FAILED=1
find "$OUT" -name "*.csv" -size +1M -delete
exit $FAILED
