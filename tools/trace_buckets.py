#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid size): which multigrid level / which slab a launch of a
shared kernel belongs to shows in its grid.  usage: trace_buckets.py <dir-or-csv> [top]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    total = 0.0
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"\(.*$", "", row["Kernel_Name"])
                name = re.sub(r"^void ", "", name)
                grid = int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0)
                wg = int(row.get("Workgroup_Size_X", row.get("Workgroup_Size", 1)) or 1)
                dt = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
                k = (name, grid // max(wg, 1))
                agg[k][0] += 1
                agg[k][1] += dt
                total += dt
    print(f"total kernel time {total:.1f} ms over {sum(v[0] for v in agg.values())} launches")
    print(f"{'kernel':60s} {'wgs':>8s} {'calls':>7s} {'ms':>10s} {'avg us':>9s} {'%':>6s}")
    for (name, wgs), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:60]:60s} {wgs:8d} {n:7d} {ms:10.2f} {1e3 * ms / n:9.1f} {100 * ms / total:6.2f}")


if __name__ == "__main__":
    main()
