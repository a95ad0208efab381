"""Round 6: the matrix-free fine level ("fine_level" 1) against the assembled one on the same state -- diagonal blocks,
residual, products, one Newmark step -- and the time of a tangent assembly either way.  python tools/r6_mf_fine_check.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg

M = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
distort = len(sys.argv) > 2 and sys.argv[2] == "d"
rng = np.random.default_rng(5)
perturb = 0.08 / n * rng.standard_normal(((n + 1) ** 3, 3)) if distort else None


def make(fine):
    G = M.Context(dim=3, degree=2, reps=(n, n, n), perturb=perturb)
    G.set_tuning("precond", 1)
    G.set_tuning("element_tangents", 2)
    G.set_tuning("fine_level", fine)
    G.set_tuning("cg_warm_start", 2)
    return G


A, B = make(0), make(1)
h = 1.0 / n
u = 0.02 * h * np.random.default_rng(1234).standard_normal(A.n)
u[A.constrained] = 0
acc = np.random.default_rng(7).standard_normal(A.n)
for G in (A, B):
    G.set(M.V_U, u)
    G.set(M.V_A_OLD, acc)
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.newton_begin_step()
    G.update_acceleration()
ra, rb = A.assemble(), B.assemble()
print("residual norm", ra, rb, abs(ra / rb - 1))
rhs_a, rhs_b = A.get(M.V_RHS), B.get(M.V_RHS)
print("rhs rel diff", np.abs(rhs_a - rhs_b).max() / np.abs(rhs_a).max())
Da, Db = A.diagonal_blocks(), B.diagonal_blocks()
print("diag blocks rel diff", np.abs(Da - Db).max() / np.abs(Da).max(), "max", np.abs(Da).max())
x = np.cos(0.37 * np.arange(A.n) + 0.11)
ya, yb = A.spmv(x), B.spmv(x)
print("spmv rel diff", np.abs(ya - yb).max() / np.abs(ya).max())
for G, name in ((A, "assembled"), (B, "matrix-free")):
    rc, its, res = G.cg_solve(1e-10, 2 * G.n)
    print(name, "cg rc", rc, "its", its, "res", res)
da, db = A.get(M.V_NEWTON), B.get(M.V_NEWTON)
print("newton update rel diff", np.abs(da - db).max() / np.abs(da).max())
for G, name in ((A, "assembled"), (B, "matrix-free")):
    t = G.bench_assemble(5)
    print(name, "ms per tangent assembly", t)
    print(name, "ms per product", G.bench_spmv(20))
A.close(); B.close()
# whole steps
for fine in (0, 1):
    G = make(fine)
    its = []
    t0 = time.time()
    for s in range(4):
        G.set_interface_traction((0.0, -2e3 * (s + 1) / 10, 0.0))
        rc, info = G.newmark_step(tol_lin=1e-6)
        its.append((info.newton_iterations, info.lin_its_total))
    print("fine_level", fine, "steps", its, "s", time.time() - t0)
    uu = G.get(M.V_U)
    if fine == 0:
        u0 = uu
    else:
        print("u rel diff after 4 steps", np.abs(uu - u0).max() / np.abs(u0).max())
    G.close()
