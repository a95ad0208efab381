"""Banded Cholesky on the device against the PCG at 1e-13 over meshes whose half bandwidths straddle the limit of the
LDS-window kernel (112 dofs): the window wraps, short last blocks, panels shorter than a block.
  python tools/direct_solver_bandwidth_sweep.py"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
M = importlib.import_module("dealii-adapter_amd")
L = M.lib()
L.mi_direct_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
worst = 0.0
for dim, p, reps in [(2, 3, (6, 5)), (2, 3, (18, 3)), (2, 4, (5, 3)), (2, 2, (40, 9)), (2, 1, (60, 25)), (2, 3, (3, 1)), (2, 1, (2, 2)),
                     (3, 1, (9, 3, 2)), (3, 1, (5, 4, 4)), (2, 2, (7, 13)), (2, 3, (4, 7)), (2, 1, (100, 54)), (2, 1, (30, 55))]:
    hi = tuple(0.1 * r for r in reps)
    G = M.Context(dim=dim, degree=p, reps=reps, hi=hi)
    G.set_tuning("precond", 0)
    rng = np.random.default_rng(3)
    G.set(M.V_U, 0.01 * 0.1 / p * rng.standard_normal(G.n) * ~G.constrained)
    G.set_interface_traction(tuple([0.0, -2e3, 0.0][:dim]))
    G.update_acceleration()
    G.assemble()
    res = C.c_double(0)
    rc = L.mi_direct_solve(G.h, C.byref(res))
    if rc != 0:
        print("dim %d p %d reps %s: %d dofs: direct solver refused (%s)" % (dim, p, reps, G.n, L.mi_last_error(G.h).decode()[:60]))
        continue
    xd = G.get(M.V_NEWTON)
    G.set(M.V_NEWTON, np.zeros(G.n))
    rc, its, _ = G.cg_solve(1e-13, 50 * G.n)
    xc = G.get(M.V_NEWTON)
    err = np.abs(xd - xc).max() / np.abs(xc).max()
    worst = max(worst, err)
    print("dim %d p %d reps %-12s %6d dofs: direct vs PCG(1e-13, %d its) rel diff %.2e" % (dim, p, reps, G.n, its, err), flush=True)
    G.close()
print("worst %.2e" % worst)
assert worst < 1e-8
