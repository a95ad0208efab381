#!/usr/bin/env python3
"""Offline prototype (container only, numpy/scipy on the oracle's matrices): which fine-level smoother gives the fewest
fine-level matrix passes per CG solve?  Two-level method = fine smoother + EXACT solve of the Q1 problem on the same
cells (on the GPU the Q2 smoother, not the coarse correction, limits the rate -- DESIGN.md section 3), so the iteration
counts transfer.  Cost model: one pass = one read of the fine matrix (a multicolour sweep reads every row once).

  python tools/proto/mg_smoothers.py [cells]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
O.lib().orc_set_threads(8)


def problem(p):
    P = O.Problem(O.make_desc(dim=3, degree=p, reps=(n, n, n)))
    P.set_interface_traction((0.0, -2e2, 0.0))
    P.update_acceleration()
    P.assemble()
    return P


t0 = time.time()
P2, P1 = problem(2), problem(1)
A, b = P2.csr().tocsr(), P2.vec(O.V_RHS).copy()
A1 = P1.csr().tocsc()
N = A.shape[0]
print("Q2 %d dofs, nnz %d; Q1 %d dofs  (%.1fs)" % (N, A.nnz, A1.shape[0], time.time() - t0))

# prolongation Q1 -> Q2 on the (2n+1)^3 lattice: even nodes coincide, odd nodes are midpoints
m2, m1 = 2 * n + 1, n + 1
I1 = sp.lil_matrix((m2, m1))
for i in range(m2):
    if i % 2 == 0:
        I1[i, i // 2] = 1.0
    else:
        I1[i, i // 2] = 0.5
        I1[i, i // 2 + 1] = 0.5
I1 = I1.tocsr()
Pn = sp.kron(I1, sp.kron(I1, I1))  # z slowest ... x fastest (node = x + m*(y + m*z))
Pr = sp.kron(Pn, sp.identity(3)).tocsr()
c2, c1 = P2.constrained, P1.constrained
Pr = sp.diags((~c2).astype(float)) @ Pr @ sp.diags((~c1).astype(float))
lu1 = spla.splu(A1)

# block-Jacobi diagonal (3x3 node blocks)
nn = N // 3
Ad = A.todia() if False else None
blocks = np.zeros((nn, 3, 3))
Acsr = A.tocsr()
for i in range(3):
    for j in range(3):
        blocks[:, i, j] = Acsr[np.arange(i, N, 3), np.arange(j, N, 3)].A1 if hasattr(Acsr[np.arange(i, N, 3), np.arange(j, N, 3)], "A1") else np.asarray(Acsr[np.arange(i, N, 3), np.arange(j, N, 3)]).ravel()
Binv = np.linalg.inv(blocks)


def dinv_apply(r):
    return np.einsum("nij,nj->ni", Binv, r.reshape(nn, 3)).reshape(-1)


# lambda_max(D^-1 A)
v = np.random.default_rng(0).standard_normal(N) * (~c2)
for _ in range(30):
    w = dinv_apply(A @ v)
    lam = np.linalg.norm(w) / np.linalg.norm(v)
    v = w / np.linalg.norm(w)
lmax = 1.15 * lam


def cheb(x, rhs, k, zero_start, ratio=20.0):
    """k Chebyshev steps on D^-1 A over [lmax/ratio, lmax]; returns x and the number of matrix passes"""
    bb, aa = lmax, lmax / ratio
    theta, delta = 0.5 * (bb + aa), 0.5 * (bb - aa)
    sigma = theta / delta
    rho_old = 1.0 / sigma
    d = np.zeros(N)
    passes = 0
    for j in range(k):
        if j == 0:
            c1_, c2_ = 0.0, 1.0 / theta
        else:
            rho = 1.0 / (2 * sigma - rho_old)
            c1_, c2_ = rho * rho_old, 2 * rho / delta
            rho_old = rho
        if j == 0 and zero_start:
            res = rhs
        else:
            res = rhs - A @ x
            passes += 1
        d = c1_ * d + c2_ * dinv_apply(res)
        x = x + d
    return x, passes


# multicolour block Gauss-Seidel: 27 colours (lattice index mod 3 per direction): nodes of one colour never share a cell
ix = np.arange(nn) % m2
iy = (np.arange(nn) // m2) % m2
iz = np.arange(nn) // (m2 * m2)
colour = (ix % 3) + 3 * (iy % 3) + 9 * (iz % 3)
rows_of = []
for c in range(27):
    nodes = np.nonzero(colour == c)[0]
    dofs = (nodes[:, None] * 3 + np.arange(3)[None, :]).reshape(-1)
    rows_of.append((nodes, dofs, Acsr[dofs, :]))


def gs_sweep(x, rhs, order):
    """one multicolour block-GS sweep = one pass over the matrix"""
    x = x.copy()
    for c in order:
        nodes, dofs, Ac = rows_of[c]
        r = rhs[dofs] - Ac @ x
        x[dofs] += np.einsum("nij,nj->ni", Binv[nodes], r.reshape(-1, 3)).reshape(-1)
    return x


FWD, BWD = list(range(27)), list(range(26, -1, -1))


def make_vcycle(kind, k):
    def apply(r):
        passes = 0
        if kind == "cheb":
            x, p = cheb(np.zeros(N), r, k, True)
            passes += p
        else:
            x = np.zeros(N)
            for _ in range(k):
                x = gs_sweep(x, r, FWD)
                passes += 1
        res = r - A @ x
        passes += 1
        xc = lu1.solve(Pr.T @ res)
        x = x + Pr @ xc
        if kind == "cheb":
            x, p = cheb(x, r, k, False)
            passes += p
        else:
            for _ in range(k):
                x = gs_sweep(x, r, BWD)
                passes += 1
        apply.passes += passes
        return x
    apply.passes = 0
    return apply


def pcg(M, tol=1e-6):
    x = np.zeros(N)
    r = b.copy()
    z = M(r)
    p = z.copy()
    rz = r @ z
    bn = np.linalg.norm(b)
    it = 0
    while np.linalg.norm(r) > tol * bn and it < 500:
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        it += 1
        if np.linalg.norm(r) <= tol * bn:
            break
        z = M(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return it


for kind, k in (("cheb", 1), ("cheb", 2), ("cheb", 3), ("gs", 1), ("gs", 2)):
    M = make_vcycle(kind, k)
    t0 = time.time()
    its = pcg(M)
    total = M.passes + its
    print("%-5s k=%d: %3d CG iterations, %4d fine matrix passes in total (%.1f per iteration)  [%.0fs]" %
          (kind, k, its, total, total / max(its, 1), time.time() - t0), flush=True)
