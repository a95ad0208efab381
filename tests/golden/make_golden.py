#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (precice/dealii-adapter) ships no golden vectors and cannot be built here (deal.II / preCICE are
absent), so the vectors are produced by the CPU oracle.  To keep them from being a mere copy of one
implementation, EVERY fixture is also computed by the independent numpy/scipy mirror in tests/golden/mirror.py
(written from the reference's formulas, not from the oracle: dense matrices, einsum on explicit 4th-order tensors,
cofactor face normals, scipy's sparse LU), and generation aborts unless the two agree -- 1e-12 for everything
assembled (quadrature points, cells, Neumann faces with the cell-QP quirk, global tangent/residual with the
constrained-diagonal rule, the linear model's K, M and boundary-value elimination), 1e-10 for everything solved
(Newton/Newmark steps, theta steps).  Fixtures: inputs + expected outputs only.  Parity stays "unpinned" in the
sense of the task rules (no reference-held vector exists); this is the most that can be pinned here.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402

MU, NU, RHO = 0.5e6, 0.4, 1000.0


from mirror import cell, material  # noqa: E402  (the independent numpy/scipy mirror, tests/golden/mirror.py)
import mirror as Mi  # noqa: E402


# ------------------------------------------------------------------ generation
TOL_ASM, TOL_STEP = 1e-12, 1e-10


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _mesh(desc):
    return Mi.Mesh(desc.dim, desc.degree, list(desc.reps), list(desc.lo), list(desc.hi), list(desc.face_role))


def nonlinear_cases():
    """(name, oracle descriptor): shared with tests/test_golden.py"""
    return [("fsi3p1", O.scenario_desc("FSI3", 2, degree=1)), ("fsi3p2", O.scenario_desc("FSI3", 2, degree=2)),
            ("fsi3p3", O.scenario_desc("FSI3", 2, degree=3)),
            ("blk3d", O.make_desc(dim=3, degree=2, reps=(2, 2, 2), hi=(0.2, 0.2, 0.2), face_role=[1, 7, 7, 7, 8, 7],
                                  body_force=(1.0, -9.81, 0.5))),
            ("pf2dp2", O.scenario_desc("PF", 2, degree=2)),
            ("blk3dq3", O.make_desc(dim=3, degree=3, reps=(1, 1, 2), hi=(0.1, 0.1, 0.2), face_role=[1, 7, 7, 7, 8, 7],
                                    body_force=(0.0, -9.81, 0.0)))]


def linear_cases():
    return [("fsi3p3", O.scenario_desc("FSI3", 2, degree=3, theta=0.5)),
            ("blk3d", O.make_desc(dim=3, degree=1, reps=(3, 2, 2), hi=(0.3, 0.2, 0.2), face_role=[1, 7, 7, 7, 8, 8],
                                  body_force=(0.0, -9.81, 0.0), theta=0.6))]


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
    S = Mi.Solid(_mesh(desc), desc.mu, desc.nu, desc.rho, (0, 0), desc.beta, desc.gamma, desc.delta_t)
    ids = P.interface_nodes
    hist, disp, log = [], [], []
    for k in range(6):
        t = (5.0 * np.sin(0.9 * k), -40.0 * min(1.0, (k + 1) / 4.0))
        P.set_interface_traction(t)
        S.set_traction(t)
        rc, info = P.newmark_step(O.SOLVER_DIRECT)
        mlog = S.newmark_step()
        assert rc == 0 and mlog["newton_iterations"] == info.newton_iterations
        assert _rel(P.vec(O.V_U), S.u) < TOL_STEP, ("fsi3 q2 trace", k, _rel(P.vec(O.V_U), S.u))
        hist.append(t)
        disp.append(P.vec(O.V_U).reshape(-1, 2)[ids].copy())
        log.append({"newton_iterations": info.newton_iterations, "assemblies": info.assemblies,
                    "res_abs": info.res_abs, "upd_abs": info.upd_abs})
    np.savez_compressed(os.path.join(HERE, "fsi3_q2_trace.npz"), traction=np.array(hist), displacement=np.array(disp),
                        interface_xy=P.coords[ids], delta_t=desc.delta_t)
    json.dump(log, open(os.path.join(HERE, "fsi3_q2_newton_log.json"), "w"), indent=1)

    # (4) global assembly, Newmark traces and the linear model on the reference's geometry (FSI3 18x3, p = 1, 2, 3)
    #     and on a 2x2x2 Q2 block with clamp, z-clamp, interface faces and a body force
    out = {}
    for name, desc in nonlinear_cases():
        dim = desc.dim
        P = O.Problem(desc)
        m = _mesh(desc)
        assert np.allclose(m.coords, P.coords, atol=1e-15) and np.array_equal(m.constrained, P.constrained)
        assert np.array_equal(m.interface_nodes, P.interface_nodes)
        S = Mi.Solid(m, desc.mu, desc.nu, desc.rho, tuple(desc.body_force)[:dim], desc.beta, desc.gamma, desc.delta_t)
        rng = np.random.default_rng(sum(map(ord, name)))
        h = (desc.hi[0] - desc.lo[0]) / desc.reps[0] / desc.degree
        free = ~m.constrained
        st = {"u": 0.01 * h * rng.standard_normal(m.n) * free, "delta": 0.005 * h * rng.standard_normal(m.n) * free,
              "v_old": 0.1 * rng.standard_normal(m.n), "a_old": rng.standard_normal(m.n),
              "traction": 2e3 * rng.standard_normal((len(m.interface_nodes), dim)), "X": rng.standard_normal((m.n, 3))}
        P.vec(O.V_U)[:], P.vec(O.V_DELTA)[:], P.vec(O.V_V_OLD)[:], P.vec(O.V_A_OLD)[:] = st["u"], st["delta"], st["v_old"], st["a_old"]
        P.set_interface_traction(st["traction"])
        P.update_acceleration()
        P.assemble()
        S.u, S.v_old, S.a_old = st["u"].copy(), st["v_old"].copy(), st["a_old"].copy()
        S.set_traction(st["traction"])
        S.a = S.a1 * st["delta"] - S.a2 * st["v_old"] - S.a3 * st["a_old"]
        K_m, rhs_m = S.assemble(st["delta"])
        K_o = P.csr()
        assert _rel(K_o.toarray(), K_m) < TOL_ASM and _rel(P.vec(O.V_RHS), rhs_m) < TOL_ASM, name
        for k, v in st.items():
            out["asm_%s_%s" % (name, k)] = v
        out["asm_%s_rhs" % name] = P.vec(O.V_RHS).copy()
        out["asm_%s_diag" % name] = K_o.diagonal()
        out["asm_%s_KX" % name] = K_o @ st["X"]
        out["asm_%s_fro" % name] = np.sqrt((K_o.data ** 2).sum())
        out["asm_%s_res_norm" % name] = P.residual_norm()
        # Newmark trace from rest under per-node tractions
        P = O.Problem(desc)
        S = Mi.Solid(m, desc.mu, desc.nu, desc.rho, tuple(desc.body_force)[:dim], desc.beta, desc.gamma, desc.delta_t)
        base = (np.array([5.0, -40.0, 3.0])[:dim] if name.startswith("fsi3") else
                np.array([30.0, -4.0])[:dim] if name.startswith("pf") else np.array([100.0, -2e3, 50.0]))
        tr, du, logs = [], [], []
        for k in range(3):
            t = base * min(1.0, (k + 1) / 2.0) + 0.1 * np.abs(base[1]) * rng.standard_normal((len(m.interface_nodes), dim))
            P.set_interface_traction(t)
            S.set_traction(t)
            rc, info = P.newmark_step(O.SOLVER_DIRECT)
            mlog = S.newmark_step()
            assert rc == 0 and mlog["newton_iterations"] == info.newton_iterations and mlog["assemblies"] == info.assemblies
            for a, b in ((P.vec(O.V_U), S.u), (P.vec(O.V_V), S.v), (P.vec(O.V_A), S.a)):
                assert _rel(a, b) < TOL_STEP, (name, k, _rel(a, b))
            tr.append(t)
            du.append(P.vec(O.V_U).copy())
            logs.append([info.newton_iterations, info.assemblies])
        out["trace_%s_traction" % name], out["trace_%s_u" % name] = np.array(tr), np.array(du)
        out["trace_%s_v" % name], out["trace_%s_a" % name] = P.vec(O.V_V).copy(), P.vec(O.V_A).copy()
        out["trace_%s_log" % name] = np.array(logs)
        print("nonlinear case", name, "ok")
    for name, desc in linear_cases():
        dim = desc.dim
        L = O.LinearProblem(desc)
        m = _mesh(desc)
        Ml = Mi.Linear(m, desc.mu, desc.nu, desc.rho, tuple(desc.body_force)[:dim], desc.delta_t, desc.theta)
        rng = np.random.default_rng(sum(map(ord, name)))
        X = rng.standard_normal((m.n, 3))
        for which, A_m in ((0, Ml.K), (1, Ml.M)):
            assert _rel(L.matrix(which).toarray(), A_m) < TOL_ASM, (name, which)
        out["lin_%s_X" % name] = X
        out["lin_%s_KX" % name], out["lin_%s_MX" % name] = L.matrix(0) @ X, L.matrix(1) @ X
        ids = m.interface_nodes
        ts, ds, vs, flags = [], [], [], []
        for k in range(4):
            t = 100.0 * rng.standard_normal((len(ids), dim))
            consistent = k != 2  # one "Force" (conservative nodal data) step
            L.vec(O.L_STRESS)[:] = 0
            Ml.stress[:] = 0
            for c in range(dim):
                L.vec(O.L_STRESS)[ids * dim + c] = t[:, c]
                Ml.stress[ids * dim + c] = t[:, c]
            rc, _, _ = L.step(O.SOLVER_DIRECT, consistent)
            Ml.step(consistent)
            assert rc == 0
            if k == 0:
                assert _rel(L.matrix(3).toarray(), Ml.system_matrix()) < TOL_ASM, name
                out["lin_%s_SX" % name] = L.matrix(3) @ X
            assert _rel(L.vec(O.L_D), Ml.d) < TOL_STEP and _rel(L.vec(O.L_V), Ml.v) < TOL_STEP, (name, k)
            ts.append(t)
            ds.append(L.vec(O.L_D).copy())
            vs.append(L.vec(O.L_V).copy())
            flags.append(int(consistent))
        out["lin_%s_traction" % name], out["lin_%s_d" % name], out["lin_%s_v" % name] = np.array(ts), np.array(ds), np.array(vs)
        out["lin_%s_consistent" % name] = np.array(flags)
        print("linear case", name, "ok")
    np.savez_compressed(os.path.join(HERE, "global_and_steps.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
