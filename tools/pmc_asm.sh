#!/bin/bash
# Issue / LDS / wait counters of the default element kernel at the headline size: bash tools/pmc_asm.sh
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_asm; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/tune_assemble.py --cells 59 --rounds 1 --reps 1 --variants 0 > $OUT/p$i.log 2>&1 || echo "pass $i ($C) failed"
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("gpurun_out/pmc_asm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "assemble_q2sf<false" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
a["launches"] = max(len(v) for v in acc.values()) if acc else 0
print(json.dumps(a, indent=1))
open("gpurun_out/pmc_asm.json", "w").write(json.dumps(a, indent=1))
PY
find $OUT -name "*.csv" -size +5M -delete
