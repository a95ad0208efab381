#!/bin/bash
# Round 6: the committed profile set from ONE box, in one gpurun call: bash tools/profile_round6.sh [outdir]
#   kernel traces (rocprofv3 --kernel-trace --stats) of the bench command on the headline path (assembled fine level) and with the
#   fine level matrix-free, the bench lines of those processes, PMC passes over both, the plain default run (CPU baseline + live
#   counter passes), emulated slabs, the assembly A/B of profiles/r06/asm_split_ab_n59.txt (experiments build).
# Copy what is wanted from <outdir> to profiles/r06/.
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
set -u
OUT=${1:-gpurun_out/round6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
for MODE in assembled matrix-free; do
  TAG=$([ $MODE = assembled ] && echo headline || echo matrix_free_fine_level)
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
done
timeout 1500 python bench.py --steps 20 --warmup 5 2>"$OUT/bench_plain.err" | tail -1 > "$OUT/bench_plain_same_box_n59.json"
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-cells 0 --no-sides --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_matrix_free_fine_level.json"
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --no-pmc --no-sides --cells 34 2>/dev/null | tail -1 > "$OUT/bench_n34_config3.json"
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --no-pmc --no-sides --cells 34 --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_n34_config3_matrix_free_fine_level.json"
timeout 600 python bench.py --cells 120 --steps 3 --warmup 1 --cpu-cells 0 --no-pmc --no-sides 2>/dev/null | tail -1 > "$OUT/bench_n120_42M_dofs.json"
timeout 600 python bench.py --cells 120 --steps 3 --warmup 1 --cpu-cells 0 --no-pmc --no-sides --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/bench_n120_42M_dofs_matrix_free_fine_level.json"
mkdir -p "$OUT/emulated_slabs"
for N in 1 2 4 8; do
  timeout 900 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs $N 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs$N.json"
  timeout 900 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs $N --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs${N}_matrix_free_fine_level.json"
done
timeout 900 python bench.py --steps 4 --warmup 2 --cpu-cells 0 --slabs 8 --scaling weak 2>/dev/null | tail -1 > "$OUT/emulated_slabs/weak_slabs8.json"
timeout 900 python bench.py --steps 4 --warmup 2 --cpu-cells 0 --slabs 8 --scaling weak --fine-level matrix-free 2>/dev/null | tail -1 > "$OUT/emulated_slabs/weak_slabs8_matrix_free_fine_level.json"
timeout 600 python tools/r6_mf_fine_check.py 59 > "$OUT/matrix_free_fine_level_check_n59.txt" 2>&1
timeout 600 python tools/r6_quad3_check.py 59 > "$OUT/smoother_quadrature_check_n59.txt" 2>&1
timeout 600 python tools/r6_quad3_check.py 24 d > "$OUT/smoother_quadrature_check_n24_distorted.txt" 2>&1
PMC_SCRIPT="tools/r6_quad3_check.py 59" timeout 900 bash tools/pmc_mf.sh "mf_spmv27" > /dev/null 2>&1; cp gpurun_out/pmc_mf.json "$OUT/pmc_counters_mf_spmv27_n59.json"
PMC_SCRIPT="tools/r6_mf_fine_check.py 59" timeout 900 bash tools/pmc_mf.sh "mf_diag<" > /dev/null 2>&1; cp gpurun_out/pmc_mf.json "$OUT/pmc_counters_mf_diag_n59.json"
timeout 900 python tools/long_run_policies.py 59 60 > "$OUT/long_run_60_steps_headline.txt" 2>&1
FINE=1 timeout 900 python tools/long_run_policies.py 59 60 > "$OUT/long_run_60_steps_matrix_free_fine_level.txt" 2>&1
timeout 900 python tools/rank_share.py > "$OUT/rank_share_n59.txt" 2>&1
if [ -f dealii-adapter_amd/libmi_elasticity_exp.so ]; then
  MI_LIB=$PWD/dealii-adapter_amd/libmi_elasticity_exp.so MI_ASM_STAMPS=1 timeout 600 python tools/r6_asm_split.py 59 > "$OUT/asm_split_stamps_n59.txt" 2>&1
fi
ls -la "$OUT"
