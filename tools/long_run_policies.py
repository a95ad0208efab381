#!/usr/bin/env python3
"""Iteration counts of a long run of the headline problem under the solver policies the executable sets (start vector from
the previous step, coarse operators rebuilt every 8th step or on demand): python tools/long_run_policies.py [cells] [steps]
(MODE=0|1|2|3 in the environment selects another "cg_warm_start"; FINE=1 the matrix-free fine level with "mf_diag_lag" 1, round 6)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _pkg  # noqa: E402

M = _pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_tuning("cg_warm_start", int(os.environ.get("MODE", "2")))
if os.environ.get("FINE", "0") == "1":
    G.set_tuning("fine_level", 1)
    G.set_tuning("mf_diag_lag", 1)
G.reset_timings()
import time
t0 = time.perf_counter()
for k in range(steps):
    ramp = min(1.0, (k + 1) / 10.0)
    G.set_interface_traction((0.0, -2e3 * ramp, 0.0))
    rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0)
    its = [int(info.lin_its[i]) for i in range(info.newton_iterations)]
    print("step %3d  newton %d  cg %s  coarse-operator rebuilds so far %d" % (k + 1, info.newton_iterations, its,
                                                                            G.get_tuning("count_mg_refresh")), flush=True)
G.get_interface_displacement()
print("%.2f ms per step over %d steps" % (1e3 * (time.perf_counter() - t0) / steps, steps))
u = G.get(M.V_U)
print("|u|_inf %.12e  sum %.12e" % (abs(u).max(), u.sum()))
