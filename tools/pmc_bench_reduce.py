#!/usr/bin/env python3
"""Reduce the per-counter rocprofv3 --pmc passes of tools/pmc_bench.sh to HBM bytes per launch for every kernel of the
bench run that moves more than 0.1 GB per launch.  gfx950 corrections (MI355X guide; checked in round 1 on the streaming
calibration kernel): read bytes = TCC_EA0_RDREQ_sum x 128 B - TCC_EA0_RDREQ_32B_sum x 96 B, write bytes = WRITE_SIZE x 1 KiB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", row["Kernel_Name"]))
            key = "%s grid=%s" % (name, row.get("Grid_Size", "?"))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    rd = c.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0 - c.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 96.0
    wr = c.get("WRITE_SIZE", 0.0) * 1024.0
    if rd + wr > 0.1e9:
        out[k] = {"launches": max(len(v) for v in cs.values()), "read_GB": rd / 1e9, "write_GB": wr / 1e9,
                  "traffic_GB_per_launch": (rd + wr) / 1e9}
print(json.dumps(out, indent=1))
