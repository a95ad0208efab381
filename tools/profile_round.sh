#!/bin/bash
# The committed profile set of a round from ONE box, in one gpurun call: bash tools/profile_round.sh [outdir]
# (rocprofv3 kernel trace of the bench command, the bench line of that process, PMC passes over the same command, the
# plain default run with the CPU baseline, the A/B runs).  Copy what is wanted from <outdir> to profiles/<round>/.
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
set -u
OUT=${1:-gpurun_out/round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --cpu-cells 0 \
  > "$OUT/bench_under_rocprof_n59.json" 2> "$OUT/trace.err"
T=$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_buckets.py "$T" > "$OUT/kernel_trace_by_grid_n59.txt"
cp "$(dirname "$T")"/*kernel_stats.csv "$OUT/kernel_stats_bench_n59.csv"
cp "$(dirname "$T")"/*domain_stats.csv "$OUT/domain_stats_bench_n59.csv"
rm -rf "$OUT/trace"
timeout 1500 bash tools/pmc_bench.sh "$OUT/pmc" > /dev/null 2>&1
cp "$OUT/pmc/pmc_bench.json" "$OUT/pmc_bench_n59.json"; rm -rf "$OUT/pmc"
timeout 1500 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_n59.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --cg-start previous-update 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_reference_cg_start.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --smoother-operator element 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_element_tangent_smoother.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --smoother-operator assembled 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_assembled_smoother.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --cg-operator element 2>/dev/null | tail -1 > "$OUT/bench_plain_same_box_cg_operator_matrix_free.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --precond-storage f32 --smoother-operator assembled 2>/dev/null | tail -1 > "$OUT/bench_precond_storage_f32_option.json"
timeout 1500 python bench.py --steps 5 --warmup 2 --cpu-cells 0 --cells 34 2>/dev/null | tail -1 > "$OUT/bench_n34_config3.json"
mkdir -p "$OUT/emulated_slabs"
for N in 1 2 4 8; do
  timeout 1500 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs $N 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs$N.json"
done
for N in 2 4 8; do
  timeout 1500 python bench.py --steps 4 --warmup 2 --cpu-cells 0 --slabs $N --scaling weak 2>/dev/null | tail -1 > "$OUT/emulated_slabs/weak_slabs$N.json"
done
timeout 400 python bench.py --cells 120 --steps 3 --warmup 1 --cpu-cells 0 2>/dev/null | tail -1 > "$OUT/bench_n120_42M_dofs.json"
timeout 900 python tools/r5_asm_ab.py 59 4 3,9 > "$OUT/assembly_kernels_n59.txt" 2>&1
timeout 900 python tools/small_case_latency.py > "$OUT/small_case_latency.txt" 2>&1
timeout 300 python tools/linear_model_latency.py > "$OUT/linear_model_latency.txt" 2>&1
# the direct solver's kernels: LDS-window kernels, the general kernel everywhere, phase clocks
(timeout 100 python tools/r5_direct_far.py; MI_BAND_LDS=0 timeout 100 python tools/r5_direct_far.py
 MI_BAND_DBG=1 timeout 100 python tools/r5_direct_far.py FSI3 2>&1 | grep clocks | head -1
 MI_BAND_DBG=1 timeout 100 python tools/r5_direct_far.py "PF 3D p=2" 2>&1 | grep clocks | head -1) > "$OUT/direct_solver_kernels.txt" 2>&1
if [ -x tools/probe/fp64_issue ]; then timeout 60 tools/probe/fp64_issue > "$OUT/fp64_issue_probe.txt" 2>&1; fi
timeout 1500 bash tools/pmc_mf.sh > /dev/null 2>&1; cp gpurun_out/pmc_mf.json "$OUT/pmc_counters_mf_spmv_n59.json"
timeout 1500 bash tools/pmc_asm.sh > /dev/null 2>&1; cp gpurun_out/pmc_asm.json "$OUT/pmc_counters_assemble_q2sf_n59.json"
timeout 900 python tools/time_element_products.py 59 2,1 > "$OUT/fine_level_product_forms_n59.txt" 2>&1
ls -la "$OUT"
timeout 900 python tools/mf_stamps.py 59 > "$OUT/mf_spmv_stage_stamps_n59.txt" 2>&1
timeout 900 python tools/mf_ablate.py 59 > "$OUT/mf_spmv_ablations_n59.txt" 2>&1
# round 5: the weak case with the first coarsened level kept replicated (the distributed level's A/B), the standard CG recurrence
# on 8 slabs (the single-reduction form's A/B), one rank's share of an 8-slab run
MI_MG_DIST_NODES=2000000000 timeout 1500 python bench.py --steps 4 --warmup 2 --cpu-cells 0 --slabs 8 --scaling weak 2>/dev/null | tail -1 > "$OUT/emulated_slabs/weak_slabs8_first_coarsened_level_replicated.json"
MI_CG_SINGLE_REDUCTION=0 timeout 1500 python bench.py --steps 8 --warmup 2 --cpu-cells 0 --slabs 8 2>/dev/null | tail -1 > "$OUT/emulated_slabs/slabs8_standard_cg_recurrence.json"
timeout 1500 python tools/rank_share.py > "$OUT/rank_share_n59.txt" 2>&1
ls -la "$OUT"
