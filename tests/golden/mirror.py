"""Independent numpy/scipy mirror of the reference algorithm -- a SECOND implementation, used only by
tests/golden/make_golden.py in the build container to cross-check the C++ oracle before fixtures are written.

Written from the reference's formulas (file:line under /root/reference), not from oracle/elasticity_oracle.cpp, and
in a different style on purpose: dense matrices, einsum on explicit 4th-order tensors, faces through the cofactor
formula  n dA = det(J) J^-T N dA_ref  instead of tangent cross products, scipy's sparse LU for every solve.

  material(), cell()        compressible_neo_hook_material.h:17-138, nonlinear_elasticity.cc:872-1036
  neumann_cell()            nonlinear_elasticity.cc:791-859 incl. the cell-QP-indexed F of :825-827
  Solid.assemble()          :1044-1087 with copy_local_to_global :760-774 (distribute_local_to_global rule)
  Solid.newmark_step()      :410-499 (Newton loop), :121-144 (run), :592-622 (Newmark), :549-576 (error norms)
  Linear                    linear_elasticity.cc:248-374 (K, M, stepping matrix, body force), :378-454 (rhs, boundary
                            values), :458-521 (consistent load), :525-586 (solve, update)

Conventions taken from deal.II (recalled, not vendored -- the same caveat as in the oracle): lexicographic tensor
order (x fastest) of shape functions and quadrature points; QProjector face order in 3D: x-faces (y,z), y-faces
(z,x), z-faces (x,y); AffineConstraints::distribute_local_to_global puts |K_e(i,i)| on constrained diagonals;
MatrixTools::apply_boundary_values keeps the diagonal, zeroes row and column, sets rhs_i = 0.
Numbering: nodes lexicographic on the (p*reps+1)^dim lattice, dof = dim*node + component (the repo's numbering;
deal.II's differs, comparisons with a real reference run would match vertices by coordinate).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

CLAMPED, INTERFACE, ZCLAMP = 1, 7, 8


# ------------------------------------------------------------------ 1D pieces
def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1), 0.5 * w


def feq_nodes(p):
    if p <= 2:
        return np.linspace(0, 1, p + 1)
    c = np.zeros(p + 1)
    c[p] = 1
    r = np.polynomial.legendre.Legendre(c).deriv().roots()
    return np.concatenate([[0.0], 0.5 * (np.sort(r.real) + 1), [1.0]])


def lagrange(nodes, x):
    n = len(nodes)
    N, dN = np.ones(n), np.zeros(n)
    for a in range(n):
        for m in range(n):
            if m != a:
                N[a] *= (x - nodes[m]) / (nodes[a] - nodes[m])
        for k in range(n):
            if k != a:
                t = 1.0 / (nodes[a] - nodes[k])
                for m in range(n):
                    if m not in (a, k):
                        t *= (x - nodes[m]) / (nodes[a] - nodes[m])
                dN[a] += t
    return N, dN


def shape_at(dim, p, nodes, xi):
    """N[a], dN[a,k] of the tensor-product basis at unit point xi (a lexicographic, x fastest)"""
    one = [lagrange(nodes, x) for x in xi]
    npc = (p + 1) ** dim
    N, dN = np.ones(npc), np.ones((npc, dim))
    for a in range(npc):
        ai = [(a // (p + 1) ** d) % (p + 1) for d in range(dim)]
        for d in range(dim):
            N[a] *= one[d][0][ai[d]]
            for k in range(dim):
                dN[a, k] *= one[d][1 if d == k else 0][ai[d]]
    return N, dN


def q1_jacobian(dim, verts, xi):
    """dX/dxi of the d-linear map through the 2^dim vertices (vertex v: bit d = upper end in direction d)"""
    Jm = np.zeros((dim, dim))
    for v in range(1 << dim):
        for j in range(dim):
            g = 1.0 if (v >> j) & 1 else -1.0
            for d in range(dim):
                if d != j:
                    g *= xi[d] if (v >> d) & 1 else 1 - xi[d]
            Jm[:, j] += verts[v] * g
    return Jm


# ------------------------------------------------------------------ material + cell (nonlinear)
def material(dim, mu, nu, F):
    kappa = 2 * mu * (1 + nu) / (3 * (1 - 2 * nu))
    I = np.eye(dim)
    J = np.linalg.det(F)
    bbar = J ** (-2.0 / dim) * F @ F.T
    IxI = np.einsum("ij,kl->ijkl", I, I)
    S = 0.5 * (np.einsum("ik,jl->ijkl", I, I) + np.einsum("il,jk->ijkl", I, I))
    devP = S - IxI / dim
    tau_bar = mu * bbar
    tau_iso = np.einsum("ijkl,kl->ij", devP, tau_bar)
    p = kappa / 2 * (J - 1 / J)
    tau = p * J * I + tau_iso
    d2 = kappa / 2 * (1 + 1 / J**2)
    Jc_vol = J * ((p + J * d2) * IxI - 2 * p * S)
    Jc_iso = (2 / dim) * np.trace(tau_bar) * devP - (2 / dim) * (np.einsum("ij,kl->ijkl", tau_iso, I) +
                                                               np.einsum("ij,kl->ijkl", I, tau_iso))
    return tau, Jc_vol + Jc_iso


def cell(dim, p, verts, u, acc, mu, nu, rho, alpha1, body):
    nodes = feq_nodes(p)
    qx, qw = gauss01(p + 2)
    npc = (p + 1) ** dim
    Ke = np.zeros((npc * dim, npc * dim))
    re = np.zeros(npc * dim)
    U = u.reshape(npc, dim)
    A = acc.reshape(npc, dim)
    for q in np.ndindex(*([p + 2] * dim)):
        qi = q[::-1]  # x fastest
        xi = np.array([qx[k] for k in qi])
        w = np.prod([qw[k] for k in qi])
        N, dN = shape_at(dim, p, nodes, xi)
        Jm = q1_jacobian(dim, verts, xi)
        G = dN @ np.linalg.inv(Jm)  # reference-configuration gradients
        JxW = np.linalg.det(Jm) * w
        F = np.eye(dim) + U.T @ G
        Fi = np.linalg.inv(F)
        g = G @ Fi  # spatial gradients
        tau, Jc = material(dim, mu, nu, F)
        a_q = A.T @ N
        grad = np.zeros((npc * dim, dim, dim))  # dof i = (a, c): grad = e_c (x) g_a
        for a in range(npc):
            for c in range(dim):
                grad[a * dim + c, c, :] = g[a]
        sym = 0.5 * (grad + grad.transpose(0, 2, 1))
        Ke += np.einsum("iab,abcd,jcd->ij", sym, Jc, sym) * JxW
        for c in range(dim):
            sl = slice(c, None, dim)
            Ke[sl, sl] += (g @ tau @ g.T + rho * alpha1 * np.outer(N, N)) * JxW
        re -= np.einsum("iab,ab->i", sym, tau) * JxW
        for c in range(dim):
            re[c::dim] -= (rho * N * (a_q[c] - body[c])) * JxW
    return Ke, re


def face_points(dim, p, nq1, face):
    """unit-cell points and weights of QGauss<dim-1>(nq1) projected to `face` (x-,x+,y-,y+,z-,z+) in QProjector order"""
    qx, qw = gauss01(nq1)
    nd, side = face // 2, face % 2
    pts, wts = [], []
    if dim == 2:
        ax = 1 if nd == 0 else 0
        for f1 in range(nq1):
            xi = np.zeros(2)
            xi[nd], xi[ax] = float(side), qx[f1]
            pts.append(xi)
            wts.append(qw[f1])
    else:
        ax0, ax1 = {0: (1, 2), 1: (2, 0), 2: (0, 1)}[nd]
        for f2 in range(nq1):
            for f1 in range(nq1):  # first face coordinate fastest
                xi = np.zeros(3)
                xi[nd], xi[ax0], xi[ax1] = float(side), qx[f1], qx[f2]
                pts.append(xi)
                wts.append(qw[f1] * qw[f2])
    return pts, wts


def neumann_cell(dim, p, verts, u_total, traction, faces, pull_back=True):
    """rhs contribution of the interface faces of one cell (:791-859).  u_total, traction: (npc, dim) nodal values.
    pull_back: the nonlinear solver's area ratio |det F F^-T N| with F taken at the CELL quadrature point whose index
    equals the face-point counter (:825-827 against :902-903); False: linear model (no pull-back)"""
    nodes = feq_nodes(p)
    nq1 = p + 2 if pull_back else p + 1
    qx, _ = gauss01(nq1)
    npc = (p + 1) ** dim
    re = np.zeros(npc * dim)
    for face in faces:
        nd, side = face // 2, face % 2
        pts, wts = face_points(dim, p, nq1, face)
        for fq, (xi_f, w) in enumerate(zip(pts, wts)):
            N, _ = shape_at(dim, p, nodes, xi_f)
            Jm = q1_jacobian(dim, verts, xi_f)
            n_ref = np.zeros(dim)
            n_ref[nd] = 1.0 if side else -1.0
            cof = np.linalg.det(Jm) * np.linalg.inv(Jm).T @ n_ref  # n dA = det J J^-T N dA_ref
            dA = np.linalg.norm(cof) * w
            normal = cof / np.linalg.norm(cof)
            scale = 1.0
            if pull_back:
                qi = [(fq // nq1**d) % nq1 for d in range(dim)]  # cell quadrature point number fq, x fastest
                xi_c = np.array([qx[k] for k in qi])
                _, dN = shape_at(dim, p, nodes, xi_c)
                G = dN @ np.linalg.inv(q1_jacobian(dim, verts, xi_c))
                F = np.eye(dim) + u_total.T @ G
                scale = np.linalg.norm(np.linalg.det(F) * np.linalg.inv(F).T @ normal)
            t_q = traction.T @ N
            re += np.outer(N, t_q * scale * dA).reshape(-1)
    return re


# ------------------------------------------------------------------ box mesh
class Mesh:
    def __init__(self, dim, p, reps, lo, hi, roles):
        self.dim, self.p = dim, p
        self.reps = list(reps)[:dim]
        self.lo, self.hi = np.array(lo[:dim], float), np.array(hi[:dim], float)
        self.roles = list(roles)
        self.nn = [p * r + 1 for r in self.reps]
        self.nnodes = int(np.prod(self.nn))
        self.n = self.nnodes * dim
        self.npc = (p + 1) ** dim
        nodes1 = feq_nodes(p)
        h = (self.hi - self.lo) / np.array(self.reps)
        self.cells = []  # (conn[npc], verts[2^dim, dim], boundary faces {face: role})
        self.coords = np.zeros((self.nnodes, dim))
        constrained = np.zeros(self.n, bool)
        on_iface = np.zeros(self.nnodes, bool)
        for c in np.ndindex(*self.reps[::-1]):
            ci = c[::-1]
            verts = np.array([[self.lo[d] + h[d] * (ci[d] + ((v >> d) & 1)) for d in range(dim)] for v in range(1 << dim)])
            conn = np.zeros(self.npc, int)
            for a in range(self.npc):
                ai = [(a // (p + 1) ** d) % (p + 1) for d in range(dim)]
                idx = [ci[d] * p + ai[d] for d in range(dim)]
                node = 0
                for d in reversed(range(dim)):
                    node = node * self.nn[d] + idx[d]
                conn[a] = node
                self.coords[node] = [self.lo[d] + h[d] * (ci[d] + nodes1[ai[d]]) for d in range(dim)]
            bfaces = {}
            for f in range(2 * dim):
                d, side = f // 2, f % 2
                if ci[d] == (self.reps[d] - 1 if side else 0) and self.roles[f]:
                    role = self.roles[f]
                    bfaces[f] = role
                    for a in range(self.npc):
                        if (a // (p + 1) ** d) % (p + 1) == (p if side else 0):
                            if role == CLAMPED:
                                constrained[conn[a] * dim:(conn[a] + 1) * dim] = True
                            elif role == ZCLAMP and dim == 3:
                                constrained[conn[a] * dim + 2] = True
                            elif role == INTERFACE:
                                on_iface[conn[a]] = True
            self.cells.append((conn, verts, bfaces))
        self.constrained = constrained
        self.interface_nodes = np.nonzero(on_iface)[0]

    def dofs(self, conn):
        return (conn[:, None] * self.dim + np.arange(self.dim)[None, :]).reshape(-1)


# ------------------------------------------------------------------ nonlinear solver
class Solid:
    def __init__(self, mesh, mu=0.5e6, nu=0.4, rho=1000.0, body=(0, 0, 0), beta=0.25, gamma=0.5, dt=0.005):
        self.m, self.mu, self.nu, self.rho, self.body = mesh, mu, nu, rho, tuple(body)
        # nonlinear_elasticity.h:242-250
        self.a1 = 1.0 / (beta * dt**2)
        self.a2 = 1.0 / (beta * dt)
        self.a3 = (1 - 2 * beta) / (2 * beta)
        self.a4 = gamma / (beta * dt)
        self.a5 = 1 - gamma / beta
        self.a6 = (1 - gamma / (2 * beta)) * dt
        n = mesh.n
        self.u, self.u_old, self.v, self.v_old, self.a, self.a_old = (np.zeros(n) for _ in range(6))
        self.stress = np.zeros(n)

    def set_traction(self, t):
        ids, dim = self.m.interface_nodes, self.m.dim
        t = np.broadcast_to(np.asarray(t, float), (len(ids), dim))
        self.stress[:] = 0
        for c in range(dim):
            self.stress[ids * dim + c] = t[:, c]

    def assemble(self, delta):
        m, dim = self.m, self.m.dim
        K, rhs = np.zeros((m.n, m.n)), np.zeros(m.n)
        ut = self.u + delta  # get_total_solution :580-588
        for conn, verts, bfaces in m.cells:
            d = m.dofs(conn)
            Ke, re = cell(dim, m.p, verts, ut[d], self.a[d], self.mu, self.nu, self.rho, self.a1, self.body)
            ifaces = [f for f, role in bfaces.items() if role == INTERFACE]
            if ifaces:
                re = re + neumann_cell(dim, m.p, verts, ut[d].reshape(-1, dim), self.stress[d].reshape(-1, dim), ifaces)
            # AffineConstraints::distribute_local_to_global with homogeneous constraints (:769-773)
            con = m.constrained[d]
            free = ~con
            K[np.ix_(d[free], d[free])] += Ke[np.ix_(free, free)]
            rhs[d[free]] += re[free]
            diag = np.abs(np.diag(Ke))
            avg = diag.mean()
            for i in np.nonzero(con)[0]:
                K[d[i], d[i]] += diag[i] if diag[i] != 0 else avg
        return K, rhs

    def newmark_step(self, max_nr=10, tol_f=1e-9, tol_u=1e-6):
        m = self.m
        free = ~m.constrained
        delta = np.zeros(m.n)
        upd = np.zeros(m.n)
        e_res = e_res0 = e_resn = e_upd = e_upd0 = e_updn = 1.0
        log = {"newton_iterations": 0, "assemblies": 0}
        it = 0
        while it < max_nr:
            self.a = self.a1 * delta - self.a2 * self.v_old - self.a3 * self.a_old  # :444
            K, rhs = self.assemble(delta)
            log["assemblies"] += 1
            e_res = np.linalg.norm(rhs[free])  # :549-560
            if it == 0:
                e_res0 = e_res
            e_resn = e_res / e_res0 if e_res0 != 0 else e_res
            if it > 0 and (e_updn <= tol_u or e_upd <= 1e-15) and (e_resn <= tol_f or e_res <= 5e-9):
                break
            upd = spla.spsolve(sp.csc_matrix(K), rhs)
            upd[m.constrained] = 0.0  # constraints.distribute :1208
            log["newton_iterations"] += 1
            e_upd = np.linalg.norm(upd[free])  # :564-576
            if it == 0:
                e_upd0 = e_upd
            e_updn = e_upd / e_upd0 if e_upd0 != 0 else e_upd
            delta = delta + upd
            it += 1
        if it >= max_nr:
            raise RuntimeError("No convergence in nonlinear solver!")
        log.update(res_abs=e_res, upd_abs=e_upd)
        self.u = self.u + delta  # :139
        self.a = self.a1 * delta - self.a2 * self.v_old - self.a3 * self.a_old
        self.v = self.a4 * delta + self.a5 * self.v_old + self.a6 * self.a_old
        self.u_old, self.v_old, self.a_old = self.u.copy(), self.v.copy(), self.a.copy()
        return log


# ------------------------------------------------------------------ linear model
class Linear:
    def __init__(self, mesh, mu=0.5e6, nu=0.4, rho=1000.0, body=(0, 0, 0), dt=0.005, theta=0.5):
        self.m, self.dt, self.theta = mesh, dt, theta
        m, dim, p = mesh, mesh.dim, mesh.p
        lam = 2 * mu * nu / (1 - 2 * nu)  # parameters.cc:189
        nodes = feq_nodes(p)
        qx, qw = gauss01(p + 1)  # quad_order = degree + 1, linear_elasticity.cc:61
        K, M = np.zeros((m.n, m.n)), np.zeros((m.n, m.n))
        self.body_vec = np.zeros(m.n)
        for conn, verts, _ in m.cells:
            d = m.dofs(conn)
            Ke, Me, be = np.zeros((len(d), len(d))), np.zeros((len(d), len(d))), np.zeros(len(d))
            for q in np.ndindex(*([p + 1] * dim)):
                qi = q[::-1]
                xi = np.array([qx[k] for k in qi])
                w = np.prod([qw[k] for k in qi])
                N, dN = shape_at(dim, p, nodes, xi)
                Jm = q1_jacobian(dim, verts, xi)
                G = dN @ np.linalg.inv(Jm)
                JxW = np.linalg.det(Jm) * w
                # :301-320   lam d_ci N_i d_cj N_j + mu d_cj N_i d_ci N_j + delta_cicj mu grad N_i . grad N_j
                GG = G @ G.T
                for ci in range(dim):
                    for cj in range(dim):
                        blk = lam * np.outer(G[:, ci], G[:, cj]) + mu * np.outer(G[:, cj], G[:, ci])
                        if ci == cj:
                            blk = blk + mu * GG
                        Ke[ci::dim, cj::dim] += blk * JxW
                    Me[ci::dim, ci::dim] += rho * np.outer(N, N) * JxW  # create_mass_matrix(rho) :341-345
                    be[ci::dim] += rho * body[ci] * N * JxW  # create_right_hand_side :358-373
            K[np.ix_(d, d)] += Ke
            M[np.ix_(d, d)] += Me
            self.body_vec[d] += be
        self.K, self.M = K, M
        self.stepping = M + dt * dt * theta * theta * K  # :347-353
        self.body_on = np.linalg.norm(body) > 1e-15
        n = m.n
        self.d, self.d_old, self.v, self.v_old, self.f_old, self.stress = (np.zeros(n) for _ in range(6))

    def consistent_load(self):
        m, dim = self.m, self.m.dim
        rhs = np.zeros(m.n)
        for conn, verts, bfaces in m.cells:
            ifaces = [f for f, role in bfaces.items() if role == INTERFACE]
            if ifaces:
                d = m.dofs(conn)
                rhs[d] += neumann_cell(dim, m.p, verts, None, self.stress[d].reshape(-1, dim), ifaces, pull_back=False)
        return rhs

    def system_matrix(self):
        """stepping matrix after MatrixTools::apply_boundary_values with zero values (:426-451)"""
        A = self.stepping.copy()
        c = self.m.constrained
        diag = np.diag(A).copy()
        A[c, :] = 0.0
        A[:, c] = 0.0
        A[c, c] = diag[c]
        return A

    def step(self, consistent=True):
        dt, th, c = self.dt, self.theta, self.m.constrained
        rhs = self.consistent_load() if consistent else self.stress.copy()  # :383-388
        self.v_old, self.d_old = self.v.copy(), self.d.copy()
        if self.body_on:
            rhs = rhs + self.body_vec
        f_new = rhs.copy()
        rhs = dt * th * rhs + dt * (1 - th) * self.f_old  # :405-409
        self.f_old = f_new
        rhs = rhs + self.M @ self.v_old - th * dt * dt * (1 - th) * (self.K @ self.v_old) - dt * (self.K @ self.d_old)
        A = self.system_matrix()
        rhs[c] = 0.0
        v = spla.spsolve(sp.csc_matrix(A), rhs)
        v[c] = 0.0
        self.v = v
        self.d = self.d + dt * th * self.v + dt * (1 - th) * self.v_old  # :579-586
        return rhs
