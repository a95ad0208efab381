#!/usr/bin/env python3
"""print the key numbers of bench.py JSON lines read from stdin (one per line), prefixed by optional labels"""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        if line:
            print(line)
        continue
    d = json.loads(line)
    c = d["config"]
    print("  ms/step %.1f | DoF-upd/s %.3g | CG its/step %.1f | Newton %.1f | assembly %.1f ms | CG %.1f ms | spmv %.3f ms "
          "(%.0f GB/s, %.1f%%) | %s" % (d["ms_per_step"], d["value"], c["cg_iterations_per_step"],
                                     c["newton_iterations_per_step"], c["ms_assembly_per_step"], c["ms_cg_per_step"],
                                     d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"],
                                     100 * d["roofline"]["frac"], c["decomposition"]))
