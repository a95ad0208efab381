set -u
OUT=gpurun_out/round6b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT/emulated_slabs"
MODE=matrix-free; TAG=matrix_free_fine_level
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --cpu-cells 0 \
  --no-pmc --no-sides --fine-level $MODE > "$OUT/bench_under_rocprof_${TAG}_n59.json" 2> "$OUT/trace_$TAG.err"
T=$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_buckets.py "$T" > "$OUT/kernel_trace_by_grid_${TAG}_n59.txt"
cp "$(dirname "$T")"/*kernel_stats.csv "$OUT/kernel_stats_bench_${TAG}_n59.csv"
rm -rf "$OUT/trace"
mkdir -p "$OUT/pmc_$TAG"
for C in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$TAG/$C" -- python3 bench.py --steps 1 --warmup 0 --cpu-cells 0 --no-pmc \
    --no-sides --fine-level $MODE > "$OUT/pmc_$TAG/$C.log" 2>&1
done
python3 tools/pmc_bench_reduce.py "$OUT/pmc_$TAG" > "$OUT/pmc_bench_${TAG}_n59.json"; rm -rf "$OUT/pmc_$TAG"
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-cells 0 --no-sides --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_matrix_free_fine_level.json"
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-cells 0 --no-pmc 2>/dev/null | tail -1 > "$OUT/bench_same_box_headline_and_sides.json"
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --no-pmc --no-sides --cells 34 --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_n34_config3_matrix_free_fine_level.json"
timeout 600 python bench.py --cells 120 --steps 3 --warmup 1 --cpu-cells 0 --no-pmc --no-sides --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_n120_42M_dofs_matrix_free_fine_level.json"
timeout 900 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs 1 --no-pmc --no-sides --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs1_matrix_free_fine_level.json"
timeout 900 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs 8 --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs8_matrix_free_fine_level.json"
ls -la "$OUT" "$OUT/emulated_slabs"
