#!/bin/bash
# LDS bank-conflict share of the default element kernel: bash tools/pmc_lds.sh
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (it exports GRAFT_REPO_ROOT); refusing to run from an unknown directory}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lds; rm -rf $OUT; mkdir -p $OUT
for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 tools/tune_assemble.py --cells 59 --rounds 1 --reps 1 --variants 0 > $OUT/$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "assemble_cells<3, 2" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in acc.items()}
print(a, "conflict share %.3f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]))
PY
