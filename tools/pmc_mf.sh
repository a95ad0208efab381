#!/bin/bash
# Counters of the matrix-free fine-level product (mf_spmv) at the headline size: bash tools/pmc_mf.sh [kernel-substring]
# PMC_SCRIPT="tools/r6_mf_fine_check.py 59" bash tools/pmc_mf.sh "mf_diag<"   -- another kernel of another python script
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-mf_spmv}
OUT=gpurun_out/pmc_mf; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
         "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 ${PMC_SCRIPT:-tools/time_element_products.py 59 2} > $OUT/p$i.log 2>&1 || echo "pass $i ($C) failed"
done
python3 - "$K" <<'PY'
import csv, glob, sys, json
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("gpurun_out/pmc_mf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if sys.argv[1] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
a["launches"] = max(len(v) for v in acc.values()) if acc else 0
print(json.dumps(a, indent=1))
open("gpurun_out/pmc_mf.json", "w").write(json.dumps(a, indent=1))
PY
find $OUT -name "*.csv" -size +5M -delete
