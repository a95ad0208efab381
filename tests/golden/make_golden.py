#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (precice/dealii-adapter) ships no golden vectors and cannot be built here (deal.II / preCICE are
absent), so the vectors are produced by the CPU oracle.  To keep them from being a mere copy of one
implementation, every quadrature-point and cell fixture is ALSO computed by the independent numpy mirror in this
file (einsum on explicit 4th-order tensors, written from the formulas of
compressible_neo_hook_material.h:17-138 and nonlinear_elasticity.cc:872-1036), and generation aborts unless the
two agree to 1e-12.  Fixtures: inputs + expected outputs only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

MU, NU, RHO = 0.5e6, 0.4, 1000.0


# ------------------------------------------------------------------ numpy mirror
def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1), 0.5 * w


def feq_nodes(p):
    if p <= 2:
        return np.linspace(0, 1, p + 1)
    # Gauss-Lobatto: endpoints + roots of P'_p
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
    idx = lambda a: [(a // (p + 1) ** d) % (p + 1) for d in range(dim)]
    Ke = np.zeros((npc * dim, npc * dim))
    re = np.zeros(npc * dim)
    U = u.reshape(npc, dim)
    A = acc.reshape(npc, dim)
    for q in np.ndindex(*([p + 2] * dim)):
        qi = q[::-1]  # x fastest
        xi = np.array([qx[k] for k in qi])
        w = np.prod([qw[k] for k in qi])
        one = [lagrange(nodes, x) for x in xi]
        N = np.array([np.prod([one[d][0][idx(a)[d]] for d in range(dim)]) for a in range(npc)])
        dN = np.array([[np.prod([one[d][1 if d == k else 0][idx(a)[d]] for d in range(dim)]) for k in range(dim)]
                       for a in range(npc)])
        Jm = np.zeros((dim, dim))
        for v in range(1 << dim):
            for j in range(dim):
                g = 1.0 if (v >> j) & 1 else -1.0
                for d in range(dim):
                    if d != j:
                        g *= xi[d] if (v >> d) & 1 else 1 - xi[d]
                Jm[:, j] += verts[v] * g
        G = dN @ np.linalg.inv(Jm)  # reference-configuration gradients
        JxW = np.linalg.det(Jm) * w
        F = np.eye(dim) + U.T @ G
        Fi = np.linalg.inv(F)
        g = G @ Fi  # spatial gradients
        tau, Jc = material(dim, mu, nu, F)
        a_q = A.T @ N
        # vector-valued shape functions: dof i = (a, c): grad = e_c (x) g_a
        grad = np.zeros((npc * dim, dim, dim))
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


# ------------------------------------------------------------------ generation
def main():
    rng = np.random.default_rng(20261002)
    out = {}
    # (1) quadrature-point fixtures: F -> tau, Jc
    for dim in (2, 3):
        Fs = np.eye(dim) + 0.2 * rng.standard_normal((6, dim, dim))
        taus, Jcs = [], []
        for F in Fs:
            _, tau_o, Jc_o = O.material(dim, MU, NU, F)
            tau_m, Jc_m = material(dim, MU, NU, F)
            assert np.abs(tau_o - tau_m).max() / np.abs(tau_m).max() < 1e-12
            assert np.abs(Jc_o - Jc_m).max() / np.abs(Jc_m).max() < 1e-12
            taus.append(tau_o)
            Jcs.append(Jc_o)
        out["qp%d_F" % dim], out["qp%d_tau" % dim], out["qp%d_Jc" % dim] = Fs, np.array(taus), np.array(Jcs)
    # (2) cell fixtures: (verts, u, acc) -> (Ke, re)
    for dim, p in ((2, 1), (2, 3), (3, 1), (3, 2)):
        desc = O.make_desc(dim=dim, degree=p, mu=MU, nu=NU, rho=RHO, body_force=(3.0, -9.81, 1.5 if dim == 3 else 0.0))
        alpha1 = 1.0 / (desc.beta * desc.delta_t**2)
        h = 0.05
        verts = h * (np.array([[(i >> d) & 1 for d in range(dim)] for i in range(1 << dim)], dtype=float) +
                     0.08 * rng.standard_normal((1 << dim, dim)))
        dpc = dim * (p + 1) ** dim
        u = 0.02 * h * rng.standard_normal(dpc)
        acc = rng.standard_normal(dpc)
        Ke_o, re_o = O.cell_tangent_residual(desc, verts, u, acc)
        Ke_m, re_m = cell(dim, p, verts, u, acc, MU, NU, RHO, alpha1, tuple(desc.body_force))
        assert np.abs(Ke_o - Ke_m).max() / np.abs(Ke_m).max() < 1e-12, (dim, p)
        assert np.abs(re_o - re_m).max() / np.abs(re_m).max() < 1e-12, (dim, p)
        k = "cell%d%d_" % (dim, p)
        out[k + "verts"], out[k + "u"], out[k + "acc"], out[k + "Ke"], out[k + "re"] = verts, u, acc, Ke_o, re_o
    np.savez_compressed(os.path.join(HERE, "qp_and_cell.npz"), **out)

    # (3) FSI3 trace: traction history -> interface displacement history + Newton log (2D, Q2, 6 windows)
    desc = O.scenario_desc("FSI3", 2, degree=2)
    P = O.Problem(desc)
    ids = P.interface_nodes
    hist, disp, log = [], [], []
    for k in range(6):
        t = (5.0 * np.sin(0.9 * k), -40.0 * min(1.0, (k + 1) / 4.0))
        P.set_interface_traction(t)
        rc, info = P.newmark_step(O.SOLVER_DIRECT)
        assert rc == 0
        hist.append(t)
        disp.append(P.vec(O.V_U).reshape(-1, 2)[ids].copy())
        log.append({"newton_iterations": info.newton_iterations, "assemblies": info.assemblies,
                    "res_abs": info.res_abs, "upd_abs": info.upd_abs})
    np.savez_compressed(os.path.join(HERE, "fsi3_q2_trace.npz"), traction=np.array(hist), displacement=np.array(disp),
                        interface_xy=P.coords[ids], delta_t=desc.delta_t)
    json.dump(log, open(os.path.join(HERE, "fsi3_q2_newton_log.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
