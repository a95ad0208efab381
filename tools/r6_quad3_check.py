"""Round 6: the smoother's fine-level operator with 3 x 3 x 3 Gauss points (tuning "smoother_quadrature" 3, mf_spmv27: two cells per
wave) against the 4 x 4 x 4 rule: how far apart the two operators are, the time of a product either way, iteration counts and
step times of both fine levels.  python tools/r6_quad3_check.py [n] [d = distorted cells]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg

M = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
distort = len(sys.argv) > 2 and sys.argv[2] == "d"
perturb = 0.08 / n * np.random.default_rng(5).standard_normal(((n + 1) ** 3, 3)) if distort else None


def make(points, fine=0):
    G = M.Context(dim=3, degree=2, reps=(n, n, n), perturb=perturb)
    G.set_tuning("precond", 1)
    G.set_tuning("element_tangents", 2)
    G.set_tuning("cg_warm_start", 2)
    if fine:
        G.set_tuning("fine_level", 1)
        G.set_tuning("mf_diag_lag", 1)
    G.set_tuning("smoother_quadrature", points)
    return G


G = make(3)
x = np.cos(0.37 * np.arange(G.n) + 0.11)
h = 1.0 / n
for amp in (0.0, 0.002, 0.02):
    u = amp * h * np.random.default_rng(1234).standard_normal(G.n)
    u[G.constrained] = 0
    G.set(M.V_U, u)
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.newton_begin_step()
    G.update_acceleration()
    G.assemble()
    assert G.get_tuning("smoother_quadrature_active") == 3
    G.set_tuning("spmv_as_smoother", 0)
    y4 = G.spmv(x)
    G.set_tuning("spmv_as_smoother", 1)
    y3 = G.spmv(x)
    print("random displacement %.3f h: |A' x - A x| / |A x| = %.3e" % (amp, np.abs(y3 - y4).max() / np.abs(y4).max()), flush=True)
G.set_tuning("spmv_variant", 4)
for s in (0, 1):
    G.set_tuning("spmv_as_smoother", s)
    print("product in the %s form: %.4f ms" % ("smoother's 27-point" if s else "64-point", G.bench_spmv(20)), flush=True)
G.close()
for fine in (0, 1):
    res = {}
    for points in (4, 3):
        G = make(points, fine)
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
        res[points] = G.get(M.V_U)
        print("fine level %s, smoother quadrature %d: %.2f ms per step, CG iterations per step %s" % (
            "matrix-free" if fine else "assembled", points, 1e3 * dt, its), flush=True)
        G.close()
    print("  displacement after 13 steps, 3 against 4: %.3e" % (np.abs(res[3] - res[4]).max() / np.abs(res[4]).max()))
