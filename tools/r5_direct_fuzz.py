"""Random meshes through the device direct solver (factorisation + substitution, and the linear model's substitutions
alone) against scipy's sparse LU: sizes and half bandwidths on all sides of the kernels' limits, short last block columns,
systems smaller than the window.
  python tools/r5_direct_fuzz.py [cases = 60] [seed = 1]"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _pkg  # noqa: E402

M = _pkg()
L = M.lib()
L.mi_direct_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, worstl, done, hist = 0.0, 0.0, 0, {}
while done < ncases:
    dim = int(rng.integers(2, 4))
    p = int(rng.integers(1, 4 if dim == 2 else 3))
    reps = tuple(int(rng.integers(1, 70 if dim == 2 else 14)) for _ in range(dim))
    nn = sorted(r * p + 1 for r in reps)
    stride, hb = 1, 0
    for k in range(dim):
        hb += stride * min(p, nn[k] - 1)
        stride *= nn[k]
    hbw, n = hb * dim + dim - 1, int(np.prod(nn)) * dim
    if hbw >= 512 or n * hbw * hbw > 3.0e8 or n > 40000:
        continue
    G = M.Context(dim=dim, degree=p, reps=reps, hi=tuple(0.1 * r for r in reps))
    G.set_tuning("precond", 0)
    G.set(M.V_U, 0.01 * 0.1 / p * rng.standard_normal(G.n) * ~G.constrained)
    G.set_interface_traction(tuple([0.0, -2e3, 0.0][:dim]))
    G.update_acceleration()
    G.assemble()
    res = C.c_double(0)
    assert L.mi_direct_solve(G.h, C.byref(res)) == 0, (dim, p, reps, L.mi_last_error(G.h))
    xd = G.get(M.V_NEWTON)
    K, b = G.csr(), G.get(M.V_RHS)
    xs = spla.spsolve(K.tocsc(), b)
    xs[G.constrained] = 0.0
    err = float(np.abs(xd - xs).max() / max(np.abs(xs).max(), 1e-300))
    # the linear model on the same mesh: factorisation in the first step, substitutions alone afterwards -- against the same
    # steps solved by the PCG at a tolerance where the result is solver independent
    ids = G.interface()[0]
    tr = [50.0 * rng.standard_normal((len(ids), dim)) for _ in range(3)]
    out = []
    for direct in (1, 0):
        H = M.Context(dim=dim, degree=p, reps=reps, hi=tuple(0.1 * r for r in reps))
        H.set_tuning("precond", 0)
        H.set_tuning("solver_type", direct)
        assert L.mi_linear_setup(H.h, 0.6) == 0
        for step in range(3):
            H.set_interface_traction(tr[step])
            its, r2 = C.c_int(0), C.c_double(0)
            assert L.mi_linear_step(H.h, 1, 1e-13, H.n * 20, C.byref(its), C.byref(r2)) == 0, L.mi_last_error(H.h)
        out.append((H.get(0).copy(), H.get(2).copy()))
        H.close()
    errl = max(float(np.abs(out[0][k] - out[1][k]).max() / max(np.abs(out[1][k]).max(), 1e-300)) for k in range(2))
    kind = "window" if hbw <= 112 else ("far" if hbw <= 160 else "general")
    hist[kind] = hist.get(kind, 0) + 1
    worst, worstl = max(worst, err), max(worstl, errl)
    print("dim %d p %d reps %-14s %6d dofs hbw %3d (%s): rel err vs scipy %.1e, linear model (3 steps) vs PCG %.1e" % (
        dim, p, reps, n, hbw, kind, err, errl), flush=True)
    assert err < 1e-8 and errl < 1e-6, (dim, p, reps)
    G.close()
    done += 1
print("worst %.2e (direct solve vs scipy), %.2e (linear model vs PCG) over %d cases %s" % (worst, worstl, done, hist))
