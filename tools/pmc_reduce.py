#!/usr/bin/env python3
"""Reduce the per-counter rocprofv3 --pmc passes of tools/pmc_spmv.sh to bytes per SpMV launch.

gfx950 corrections (MI355X guide, checked on the streaming-read calibration kernel, whose true byte count is known):
read bytes = TCC_EA0_RDREQ_sum x 128 B when there are no 32-byte requests (FETCH_SIZE reports half of that);
write bytes = WRITE_SIZE x 1 KiB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    root, n = sys.argv[1], int(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", row["Kernel_Name"]))
                if "sell_spmv" in name or "stream_read" in name:
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    raw = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    out = {"raw_counter_averages_per_launch": raw, "workload": "%d^3 Q2 cells" % n}
    nodes = (2 * n + 1) ** 3
    c1 = (n - 1) * 5 + 2 * 3 + n * 3
    nnzb = c1 ** 3
    for k, c in raw.items():
        rd = c.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0 - c.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 96.0
        wr = c.get("WRITE_SIZE", 0.0) * 1024.0
        entry = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
                 "FETCH_SIZE_x1024": c.get("FETCH_SIZE", 0.0) * 1024.0}
        if "sell_spmv" in k:
            icol = os.environ.get("MI_SELL_ICOL", "1") != "0"  # generated column indices: 8 B per row instead of 4 B per block
            entry["algorithmic_bytes_per_launch"] = nnzb * 72 + (nodes * 8 if icol else nnzb * 4) + nodes * 4 + nodes * 24 * 2
            entry["ratio"] = (rd + wr) / entry["algorithmic_bytes_per_launch"]
            out["kernel"] = k
            out["traffic_bytes_per_launch"] = rd + wr
            out["algorithmic_bytes_per_launch"] = entry["algorithmic_bytes_per_launch"]
        else:
            entry["true_bytes"] = nnzb * 72
        out[k] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
