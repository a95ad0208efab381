#!/bin/bash
# Hardware counters of the element kernels (default and sum-factorised), one rocprofv3 --pmc pass per counter.
#   bash tools/pmc_assemble.sh [cells]   -> gpurun_out/pmc_assemble_n<cells>.json
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
set -u
N=${1:-59}
OUT=gpurun_out/pmc_asm_n$N
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
for C in SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
         SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -- python3 tools/tune_assemble.py --cells "$N" --rounds 1 --reps 1 \
    --variants 0,5 > "$OUT/$C.log" 2>&1 || echo "pass $C failed (see $OUT/$C.log)"
done
python3 - "$OUT" "$N" > "gpurun_out/pmc_assemble_n$N.json" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
root, n = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", row["Kernel_Name"]))
            if name.startswith("mi::assemble_cells<3, 2") or name.startswith("mi::assemble_cells_sf"):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"workload": "%d^3 Q2 cells, per colour launch (8 launches per assembly)" % n, "kernels": {}}
for k, cs in acc.items():
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(avg)
    f64 = avg.get("SQ_INSTS_VALU_FMA_F64", 0) + avg.get("SQ_INSTS_VALU_MUL_F64", 0) + avg.get("SQ_INSTS_VALU_ADD_F64", 0)
    if avg.get("SQ_INSTS_VALU"):
        d["fp64_share_of_valu_instructions"] = f64 / avg["SQ_INSTS_VALU"]
    if avg.get("SQ_INSTS_LDS"):
        d["fp64_instructions_per_lds_instruction"] = f64 / avg["SQ_INSTS_LDS"]
    if avg.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_active_cycles"] = avg.get("SQ_LDS_BANK_CONFLICT", 0) / avg["SQ_LDS_IDX_ACTIVE"]
    if avg.get("SQ_BUSY_CYCLES"):
        d["valu_active_share_of_busy_cycles"] = avg.get("SQ_ACTIVE_INST_VALU", 0) / avg["SQ_BUSY_CYCLES"]
        d["lds_active_share_of_busy_cycles"] = avg.get("SQ_ACTIVE_INST_LDS", 0) / avg["SQ_BUSY_CYCLES"]
    out["kernels"][k] = d
print(json.dumps(out, indent=1))
PY
cat "gpurun_out/pmc_assemble_n$N.json"
find "$OUT" -name "*.csv" -size +5M -delete
