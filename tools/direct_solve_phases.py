import os, sys, time
sys.path.insert(0, ".")
from bench import _pkg
M = _pkg()
G = M.Context(dim=2, degree=3, reps=(18, 3), lo=(0.24899, 0.19), hi=(0.6, 0.21), face_role=[1, 7, 7, 7, 8, 8])
G.set_tuning("solver_type", 1)
G.set_interface_traction((0.0, -4.0)); G.update_acceleration(); G.assemble()
import ctypes as C
L = M.lib()
L.mi_direct_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
res = C.c_double(0)
for _ in range(5): L.mi_direct_solve(G.h, C.byref(res))
t0 = time.perf_counter()
for _ in range(50): L.mi_direct_solve(G.h, C.byref(res))
print("MI_BAND_DBG=%s: mi_direct_solve %.3f ms" % (os.environ.get("MI_BAND_DBG"), 1e3 * (time.perf_counter() - t0) / 50))
