"""Round 6: layouts of the matrix-free kernels' result slots under the 27-point smoother product (tuning "mf_slots_cell_major":
1 cell-major, 2 line-major, 0 node-major) -- the product with its gather, and steps of the ramp on both fine levels, all in
one process.  python tools/r6_slot_layout.py [n] [layouts, e.g. 1,2,1,2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg

M = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
layouts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,1,2").split(",")]
ref = {}
for fine in (0, 1):
    for lay in layouts:
        G = M.Context(dim=3, degree=2, reps=(n, n, n))
        G.set_tuning("precond", 1)
        G.set_tuning("element_tangents", 2)
        G.set_tuning("cg_warm_start", 2)
        if fine:
            G.set_tuning("fine_level", 1)
            G.set_tuning("mf_diag_lag", 1)
        G.set_tuning("mf_slots_cell_major", lay)
        its = []
        for s in range(3):
            G.set_interface_traction((0.0, -2e3 * (s + 1) / 10, 0.0))
            G.newmark_step(tol_lin=1e-6)
        G.get_interface_displacement()
        t0 = time.perf_counter()
        for s in range(3, 13):
            G.set_interface_traction((0.0, -2e3 * min(1.0, (s + 1) / 10), 0.0))
            rc, info = G.newmark_step(tol_lin=1e-6)
            its.append(info.lin_its_total)
        G.get_interface_displacement()
        dt = (time.perf_counter() - t0) / 10
        u = G.get(M.V_U)
        ref.setdefault(fine, u)
        G.set_tuning("spmv_variant", 4)
        G.set_tuning("spmv_as_smoother", 1)
        tp = G.bench_spmv(40)
        print("fine level %s, slot layout %d: %.2f ms per step, %d CG iterations in 10 steps, smoother product + gather %.4f ms, "
              "displacement against the first run %.2e" % ("matrix-free" if fine else "assembled", lay, 1e3 * dt, sum(its), tp,
                                                            np.abs(u - ref[fine]).max() / np.abs(ref[fine]).max()), flush=True)
        G.close()
