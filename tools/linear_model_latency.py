#!/usr/bin/env python3
"""Per-step wall time of the linear theta-model (BASELINE configuration 2: Q1 block 40^3 = 206,763 dofs, and the
shipped 2D FSI3 case):  python tools/linear_model_latency.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg  # noqa: E402


def main():
    M = _pkg()
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    cases = [("block 3D Q1 40^3 (config 2)", 3, 1, (40, 40, 40), (0, 0, 0), (1, 1, 1)),
             ("beam 3D Q1 80x8x8 [0,10]x[0,1]^2", 3, 1, (80, 8, 8), (0, 0, 0), (10, 1, 1)),
             ("FSI3 2D p=3 (shipped)", 2, 3, (18, 3), (0.24899, 0.19), (0.6, 0.21))]
    for name, dim, p, reps, lo, hi in cases + [(c[0] + ", Solver type = Direct",) + c[1:] for c in cases[2:]]:
        t0 = time.perf_counter()
        G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=[1, 7, 7, 7, 8, 8])
        if name.endswith("Direct"):  # banded Cholesky of the constant system matrix, factorised once
            G.set_tuning("solver_type", 1)
        assert L.mi_linear_setup(G.h, 0.5) == 0, L.mi_last_error(G.h)
        t_setup = time.perf_counter() - t0
        tr = (0.0, -200.0, 0.0)[:dim]
        its_sum, n = 0, 20
        for k in range(3 + n):
            if k == 3:
                t0 = time.perf_counter()
            G.set_interface_traction(tr)
            its, res = C.c_int(0), C.c_double(0)
            rc = L.mi_linear_step(G.h, 1, 1e-10, G.n, C.byref(its), C.byref(res))
            assert rc == 0, L.mi_last_error(G.h)
            if k >= 3:
                its_sum += its.value
        dt = (time.perf_counter() - t0) / n
        print("%-36s %8d dofs  setup %.2f s  %.2f ms/step  cg %.1f its/step" % (name, G.n, t_setup, 1e3 * dt, its_sum / n),
              flush=True)


if __name__ == "__main__":
    main()
