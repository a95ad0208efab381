#!/usr/bin/env python3
"""One-off stress of the unassembled smoother operators (2: matrix-free from the quadrature-point records, 1: stored
element tangents): random 3D Q2 meshes (cells, box, distortion or none, boundary roles, slab count, state, tractions) with
the form forced on; the product must equal the assembled product, the multigrid-PCG must give the same solution and
(+-1) iteration count as with the assembled smoother, and the sum-factorised element kernel must reproduce the node-pair
kernel's tangent and residual.
  python tools/stress_element_tangents.py [n = 24] [seed = 0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg  # noqa: E402

M = _pkg()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = [0.0, 0.0, 0.0]
for case in range(n_cases):
    reps = tuple(int(v) for v in rng.integers(3, 8, 3))
    slabs = int(rng.integers(1, min(3, reps[2]) + 1))
    hi = tuple(float(0.1 * r * rng.uniform(0.7, 1.4)) for r in reps)
    roles = [1] + [int(rng.choice([0, 7, 7, 8 if f >= 4 else 7])) for f in range(1, 6)]
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = 0.004 * rng.standard_normal((nverts, 3)) if case % 3 else None  # every third mesh: axis-parallel boxes
    res = {}
    for op in (0, 1, 2):
        G = M.Context(dim=3, degree=2, reps=reps, hi=hi, face_role=roles, perturb=perturb, slabs=slabs,
                      body_force=(0.0, -9.81, 0.0))
        G.set_tuning("precond", 1)
        G.set_tuning("mg_fuse", 0)
        if op:
            G.set_tuning("element_tangents", op)
        G.set_tuning("smoother_operator", op)
        if op == 0:
            G.set_tuning("asm_variant", 9)  # the node-pair kernel assembles the reference run
        r2 = np.random.default_rng(1000 + case)
        free = ~G.constrained
        h = min(hi[d] / reps[d] for d in range(3)) / 2
        G.set(M.V_U, 0.01 * h * r2.standard_normal(G.n) * free)
        G.set(M.V_V_OLD, 0.1 * r2.standard_normal(G.n))
        nif = len(G.interface()[0])
        if nif:
            G.set_interface_traction(2e3 * r2.standard_normal((nif, 3)))
        G.update_acceleration()
        G.assemble()
        x = r2.standard_normal(G.n)
        G.set_tuning("spmv_variant", 3)
        y3 = G.spmv(x)
        if op:
            G.set_tuning("spmv_variant", 4)
            y4 = G.spmv(x)
            worst[0] = max(worst[0], np.abs(y4 - y3).max() / np.abs(y3).max())
            G.set_tuning("spmv_variant", 3)
        rc, its, _ = G.cg_solve(rel_tol=1e-10)
        assert rc == 0, (case, op, reps, slabs)
        res[op] = (its, G.get(M.V_NEWTON), y3, G.get(M.V_RHS))
        G.close()
    d = max(np.abs(res[o][1] - res[0][1]).max() / np.abs(res[0][1]).max() for o in (1, 2))
    worst[1] = max(worst[1], d)
    # element kernels: sum-factorised (runs 1, 2) against node-pair (run 0): same product of the assembled matrix, same rhs
    dk = max(np.abs(res[2][2] - res[0][2]).max() / np.abs(res[0][2]).max(),
             np.abs(res[2][3] - res[0][3]).max() / max(np.abs(res[0][3]).max(), 1e-300))
    worst[2] = max(worst[2], dk)
    ok = all(abs(res[0][0] - res[o][0]) <= 1 for o in (1, 2)) and d < 1e-7 and dk < 1e-12
    print("case %2d reps %s slabs %d roles %s %s: its %d / %d / %d, solution diff %.1e, element kernels %.1e %s" %
          (case, reps, slabs, roles, "boxes" if perturb is None else "distorted", res[0][0], res[1][0], res[2][0], d, dk,
           "" if ok else "  <-- MISMATCH"), flush=True)
    assert ok
print("worst product diff %.2e, worst solution diff %.2e, worst element-kernel diff %.2e over %d cases" %
      (worst[0], worst[1], worst[2], n_cases))
