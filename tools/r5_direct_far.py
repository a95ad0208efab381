"""Latency of one factorisation + substitution (mi_direct_solve) on the reference's geometries, for the A/B of the banded
Cholesky kernels (MI_BAND_LDS=0: the general kernel everywhere; MI_BAND_DBG=1: phase clocks of the LDS-window kernel).
  python tools/r5_direct_far.py [substring of the case name]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _pkg  # noqa: E402

M = _pkg()
L = M.lib()
L.mi_direct_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
CASES = [("FSI3 2D p=3", 2, 3, (18, 3), (0.24899, 0.19), (0.6, 0.21)),
         ("PF 3D p=2", 3, 2, (3, 18, 1), (-0.05, 0.0, 0.0), (0.05, 1.0, 0.3)),
         ("PF 3D p=1", 3, 1, (3, 18, 1), (-0.05, 0.0, 0.0), (0.05, 1.0, 0.3)),
         ("plate 2D p=2 18x30", 2, 2, (18, 30), (0.0, 0.0), (1.8, 3.0)),
         ("plate 2D p=1 77x60", 2, 1, (77, 60), (0.0, 0.0), (7.7, 6.0))]
for name, dim, p, reps, lo, hi in CASES:
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=[1, 7, 7, 7, 8, 8])
    G.set_tuning("solver_type", 1)
    G.set_interface_traction((0.0, -4.0, 0.0)[:dim])
    G.update_acceleration()
    G.assemble()
    res = C.c_double(0)
    for _ in range(5):
        assert L.mi_direct_solve(G.h, C.byref(res)) == 0
    t0 = time.perf_counter()
    for _ in range(30):
        L.mi_direct_solve(G.h, C.byref(res))
    dt = 1e3 * (time.perf_counter() - t0) / 30
    print("MI_BAND_LDS=%s  %-20s %6d dofs: mi_direct_solve %.3f ms" % (os.environ.get("MI_BAND_LDS", "1"), name, G.n, dt), flush=True)
    G.close()
