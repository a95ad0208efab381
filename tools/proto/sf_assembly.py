"""Offline prototype of the sum-factorised element tangent (assemble_q2sf in mi_kernels.hip): the coefficient fields
C^{ij}_{kl}(q) and the three contractions x -> y -> z with the lane decomposition of the kernel (item = (ij, (a1 >= b1),
a2)), checked against the dense element tangent of the independent mirror.  3D Q2, 4^3 Gauss points."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import mirror as m

rng = np.random.default_rng(5)
dim, p = 3, 2
mu, nu, rho, alpha1 = 0.5e6, 0.4, 1000.0, 1.0 / (0.25 * 0.005 ** 2)
kappa = 2 * mu * (1 + nu) / (3 * (1 - 2 * nu))
verts = np.array([[(v >> d) & 1 for d in range(3)] for v in range(8)], float) * [0.3, 0.2, 0.25]
verts += 0.02 * rng.standard_normal(verts.shape)
u = 0.01 * rng.standard_normal((27, 3))
Ke = m.cell(dim, p, verts, u, np.zeros((27, 3)), mu, nu, rho, alpha1, np.zeros(3))[0]

nodes = m.feq_nodes(p)
qx, qw = m.gauss01(p + 2)
S = np.zeros((4, 3)); D = np.zeros((4, 3))
for q in range(4):
    S[q], D[q] = m.lagrange(nodes, qx[q])
PHI = [S, D]  # PHI[flag][q][a]

# coefficient fields C[i][j][k][l][qz][qy][qx] and the mass field
C = np.zeros((3, 3, 3, 3, 4, 4, 4)); MU = np.zeros((4, 4, 4))
for qz in range(4):
    for qy in range(4):
        for qxx in range(4):
            xi = np.array([qx[qxx], qx[qy], qx[qz]])
            Jm = m.q1_jacobian(dim, verts, xi); Ji = np.linalg.inv(Jm)
            w = np.linalg.det(Jm) * qw[qxx] * qw[qy] * qw[qz]
            dN = np.zeros((27, 3))
            for a in range(27):
                i, j, k = a % 3, (a // 3) % 3, a // 9
                dN[a] = [D[qxx, i] * S[qy, j] * S[qz, k], S[qxx, i] * D[qy, j] * S[qz, k], S[qxx, i] * S[qy, j] * D[qz, k]]
            F = np.eye(3) + (u.T @ dN) @ Ji
            J = np.linalg.det(F); b = F @ F.T
            s = mu * J ** (-2.0 / 3); tr = s * np.trace(b)
            tiso = s * b - tr / 3 * np.eye(3); tau = tiso + 0.5 * kappa * (J * J - 1) * np.eye(3)
            cII = kappa * J * J - 2.0 / 9 * tr; cs2 = 0.5 * (-kappa * (J * J - 1) + 2.0 / 3 * tr)
            M = Ji @ np.linalg.inv(F)
            Tm = -(2.0 / 3) * M @ tiso
            A = w * (cII * M + Tm); B = w * M; E = w * cs2 * M
            Sk = w * (cs2 * M @ M.T + M @ tau @ M.T)
            for i in range(3):
                for j in range(3):
                    for k in range(3):
                        for l in range(3):
                            C[i, j, k, l, qz, qy, qxx] = A[k, i] * M[l, j] + B[k, i] * Tm[l, j] + E[k, j] * M[l, i] + (Sk[k, l] if i == j else 0.0)
            MU[qz, qy, qxx] = alpha1 * rho * w

pairs = [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2)]
K = np.zeros((81, 81)); done = np.zeros((81, 81), bool)
for ij in range(9):
    i, j = ij // 3, ij % 3
    for (a1, b1) in pairs:
        for a2 in range(3):
            Kacc = np.zeros((3, 3, 3))  # [a3][b3][b2]
            for qz in range(4):
                acc = np.zeros((2, 2, 3))  # [k==z][l==z][b2]
                for kl in range(10):
                    if kl == 9:
                        if i != j:
                            continue
                        fld, kx, lx, ky, ly, kz, lz = MU, 0, 0, 0, 0, 0, 0
                    else:
                        k, l = kl // 3, kl % 3
                        fld = C[i, j, k, l]
                        kx, lx, ky, ly, kz, lz = int(k == 0), int(l == 0), int(k == 1), int(l == 1), int(k == 2), int(l == 2)
                    for qy in range(4):
                        c4 = fld[qz, qy, :]
                        t = sum(PHI[kx][q, a1] * PHI[lx][q, b1] * c4[q] for q in range(4))
                        ta = t * PHI[ky][qy, a2]
                        for b2 in range(3):
                            acc[kz, lz, b2] += ta * PHI[ly][qy, b2]
                for kz in range(2):
                    for lz in range(2):
                        for b2 in range(3):
                            v = acc[kz, lz, b2]
                            for a3 in range(3):
                                va = v * PHI[kz][qz, a3]
                                for b3 in range(3):
                                    Kacc[a3, b3, b2] += va * PHI[lz][qz, b3]
            for a3 in range(3):
                for b3 in range(3):
                    for b2 in range(3):
                        a = a1 + 3 * a2 + 9 * a3; b = b1 + 3 * b2 + 9 * b3
                        ar = a3 + 3 * a2 + 9 * a1; br = b3 + 3 * b2 + 9 * b1   # digit-reversed order decides the orientation
                        if ar < br:
                            continue
                        K[3 * a + i, 3 * b + j] = Kacc[a3, b3, b2]; done[3 * a + i, 3 * b + j] = True
                        K[3 * b + j, 3 * a + i] = Kacc[a3, b3, b2]; done[3 * b + j, 3 * a + i] = True
print("covered", done.all(), "rel err", np.abs(K - Ke).max() / np.abs(Ke).max())
