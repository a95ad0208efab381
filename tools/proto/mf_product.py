"""Offline prototype of the matrix-free element product (mf_spmv in mi_kernels.hip): y_e = K_e x_e from the
quadrature-point records M, tau, w, w*c_II, c_S/2 by sum factorisation, checked against the dense element tangent of
the independent mirror (tests/golden/mirror.py).  3D Q2, 4^3 Gauss points.  Not part of the product or the tests."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import mirror as m

rng = np.random.default_rng(3)
dim, p = 3, 2
mu, nu, rho, alpha1 = 0.5e6, 0.4, 1000.0, 1.0 / (0.25 * 0.005 ** 2)
kappa = 2 * mu * (1 + nu) / (3 * (1 - 2 * nu))
verts = np.array([[(v >> d) & 1 for d in range(3)] for v in range(8)], float) * [0.3, 0.2, 0.25]
verts += 0.02 * rng.standard_normal(verts.shape)
u = 0.01 * rng.standard_normal((27, 3))
Ke, re = m.cell(dim, p, verts, u, np.zeros((27, 3)), mu, nu, rho, alpha1, np.zeros(3))[:2]
x = rng.standard_normal((27, 3))
y_ref = (Ke @ x.reshape(-1)).reshape(27, 3)

nodes = m.feq_nodes(p)
qx, qw = m.gauss01(p + 2)
S = np.zeros((4, 3)); D = np.zeros((4, 3))
for q in range(4):
    S[q], D[q] = m.lagrange(nodes, qx[q])

# quadrature-point records (what assemble_cells phase A leaves in LDS)
rec = []
for qz in range(4):
    for qy in range(4):
        for qxx in range(4):
            xi = np.array([qx[qxx], qx[qy], qx[qz]])
            Jm = m.q1_jacobian(dim, verts, xi)
            Ji = np.linalg.inv(Jm)
            w = np.linalg.det(Jm) * qw[qxx] * qw[qy] * qw[qz]
            dN = np.zeros((27, 3))
            for a in range(27):
                i, j, k = a % 3, (a // 3) % 3, a // 9
                dN[a] = [D[qxx, i] * S[qy, j] * S[qz, k], S[qxx, i] * D[qy, j] * S[qz, k], S[qxx, i] * S[qy, j] * D[qz, k]]
            gu = (u.T @ dN) @ Ji
            F = np.eye(3) + gu
            J = np.linalg.det(F)
            b = F @ F.T
            s = mu * J ** (-2.0 / 3)
            tr = s * np.trace(b)
            tiso = s * b - tr / 3 * np.eye(3)
            tau = tiso + 0.5 * kappa * (J * J - 1) * np.eye(3)
            cII = kappa * J * J - 2.0 / 9 * tr
            cS = -kappa * (J * J - 1) + 2.0 / 3 * tr
            rec.append((Ji @ np.linalg.inv(F), tau, w, w * cII, 0.5 * cS))

# evaluate: X[c][k][j][i]
X = x.T.reshape(3, 3, 3, 3)
A_S = np.einsum("qi,ckji->ckjq", S, X); A_D = np.einsum("qi,ckji->ckjq", D, X)
B_DS = np.einsum("rj,ckjq->ckrq", S, A_D); B_SD = np.einsum("rj,ckjq->ckrq", D, A_S); B_SS = np.einsum("rj,ckjq->ckrq", S, A_S)
H = np.zeros((3, 3, 4, 4, 4)); V = np.zeros((3, 4, 4, 4))   # H[c][d][qz][qy][qx]
H[:, 0] = np.einsum("sk,ckrq->csrq", S, B_DS); H[:, 1] = np.einsum("sk,ckrq->csrq", S, B_SD)
H[:, 2] = np.einsum("sk,ckrq->csrq", D, B_SS); V = np.einsum("sk,ckrq->csrq", S, B_SS)
Q = np.zeros_like(H); Vm = np.zeros_like(V)
for q, (M, tau, w, wcII, cs2) in enumerate(rec):
    qz, qy, qxx = q // 16, (q // 4) % 4, q % 4
    h = H[:, :, qz, qy, qxx] @ M                       # h[j][k] = sum_l H[j][l] M[l][k]
    tiso = tau - np.trace(tau) / 3 * np.eye(3)
    trh = np.trace(h)
    Sm = (wcII * trh - (2.0 / 3) * w * np.sum(tiso * h)) * np.eye(3) + w * (-(2.0 / 3) * trh * tiso + cs2 * (h + h.T) + h @ tau)
    Q[:, :, qz, qy, qxx] = Sm @ M.T                    # Q[i][k] = sum_j S[i][j] M[k][j]
    Vm[:, qz, qy, qxx] = alpha1 * rho * w * V[:, qz, qy, qxx]
C_DS = np.einsum("sk,csrq->ckrq", S, Q[:, 0]); C_SD = np.einsum("sk,csrq->ckrq", S, Q[:, 1])
C_SS = np.einsum("sk,csrq->ckrq", D, Q[:, 2]) + np.einsum("sk,csrq->ckrq", S, Vm)
E_D = np.einsum("rj,ckrq->ckjq", S, C_DS); E_S = np.einsum("rj,ckrq->ckjq", D, C_SD) + np.einsum("rj,ckrq->ckjq", S, C_SS)
Y = np.einsum("qi,ckjq->ckji", D, E_D) + np.einsum("qi,ckjq->ckji", S, E_S)
y = Y.reshape(3, 27).T
print("rel err", np.abs(y - y_ref).max() / np.abs(y_ref).max())
