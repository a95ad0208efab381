#!/usr/bin/env python3
"""Per-step wall time of the reference's own small geometries (launch-latency regime, not bandwidth):
   python tools/small_case_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import _pkg  # noqa: E402


def main():
    M = _pkg()
    cases = [("FSI3 2D p=3 (shipped degree)", 2, 3, (18, 3), (0.24899, 0.19), (0.6, 0.21)),
             ("FSI3 2D p=1", 2, 1, (18, 3), (0.24899, 0.19), (0.6, 0.21)),
             ("FSI3 2D p=2", 2, 2, (18, 3), (0.24899, 0.19), (0.6, 0.21)),
             ("PF 3D p=2", 3, 2, (3, 18, 1), (-0.05, 0.0, 0.0), (0.05, 1.0, 0.3)),
             ("block 3D p=2 12^3", 3, 2, (12, 12, 12), (0, 0, 0), (1, 1, 1))]
    if len(sys.argv) > 1:  # block sizes: cells per side
        cases = [("block 3D p=2 %d^3" % n, 3, 2, (n, n, n), (0, 0, 0), (1, 1, 1)) for n in map(int, sys.argv[1:])]
    # solver: "mg" / "jacobi" = PCG at the shipped Residual 1e-6 with that preconditioner; "direct" = Solver type = Direct
    # (banded Cholesky on the device, the reference's shipped default); "pcg 1e-12" = what served Direct until round 3
    for solver in ("direct", "pcg 1e-12", "jacobi", "mg"):
        for name, dim, p, reps, lo, hi in cases:
            roles = [1, 7, 7, 7, 8, 8]
            G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles)
            G.set_tuning("precond", 1 if solver == "mg" else 0)
            G.set_tuning("solver_type", 1 if solver == "direct" else 0)
            tol = 1e-12 if solver == "pcg 1e-12" else 1e-6
            precond = solver
            t = (0.0, -40.0, 0.0)[:dim]
            its = newton = 0
            for k in range(3):
                G.set_interface_traction(tuple(min(1.0, (k + 1) / 10) * x for x in t))
                G.newmark_step(tol_lin=tol, max_it_mult=10.0)
            t0 = time.perf_counter()
            n = 10
            for k in range(3, 3 + n):
                G.set_interface_traction(tuple(min(1.0, (k + 1) / 10) * x for x in t))
                rc, info = G.newmark_step(tol_lin=tol, max_it_mult=10.0)
                assert rc == 0
                its += info.lin_its_total
                newton += info.newton_iterations
            dt = (time.perf_counter() - t0) / n
            print("%-30s solver=%-10s %6d dofs  %.2f ms/step  newton %.1f  linear iterations %.1f" %
                  (name, precond, G.n, 1e3 * dt, newton / n, its / n), flush=True)


if __name__ == "__main__":
    main()
