#!/usr/bin/env python3
"""Wall time of consecutive Newmark steps of the headline workload (is there a clock ramp?): python tools/step_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg  # noqa: E402

M = _pkg()
n = 59
G = M.Context(dim=3, degree=2, reps=(n, n, n))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for k in range(steps):
    G.set_interface_traction((0.0, -2e3 * min(1.0, (k + 1) / 10.0), 0.0))
    G.set_profiling(True)
    G.reset_timings()
    t0 = time.perf_counter()
    rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0)
    dt = time.perf_counter() - t0
    tm = G.timings()
    print("step %2d: %.1f ms  newton %d cg %d  spmv %.3f ms" % (k, 1e3 * dt, info.newton_iterations, info.lin_its_total,
                                                                 tm["spmv"][0] / max(tm["spmv"][1], 1)), flush=True)
