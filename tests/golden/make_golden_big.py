#!/usr/bin/env python3
"""Oracle results at the BASELINE sizes, kept as small fixtures (tests/golden/baseline_sizes.npz).

Until round 3 everything the oracle was compared with ran on <= ~10 k DoFs, where the product's preconditioner is
Jacobi and its smoother the assembled matrix; the default big-mesh path (multigrid + matrix-free smoother + the
sum-factorised element kernel) was only ever compared with itself.  This script runs the CPU restatement of the
reference algorithm (oracle/, reference lines cited there) at the sizes where that path is active and keeps a
lattice subsample of the results plus whole-vector functionals:

  blk24   24^3 Q2 block (117,649 nodes: the smallest mesh on which the matrix-free smoother is the default), two full
          Newmark steps, traction (0,-2e3,0), linear tolerance 1e-12       [REF nonlinear_elasticity.cc:410-499]
  blk24d  the same block with every cell distorted (vertices moved by 8 % of the cell size, seeded): the general-geometry
          branches of the element kernel and of the matrix-free product at the default solver path
  q1_48   48^3 Q1 cells (352,947 DoFs) and
  q2_2d   300^2 Q2 cells in 2D (722,402 DoFs), two steps each: the other element families on the multigrid path
  cfg3    BASELINE configuration 3: 34^3 Q2 block (985,527 DoFs), the first three Newmark steps of the bench's ramp
          (traction (0,-2e2 k,0), k = 1, 2, 3), "Residual" = 1e-10          [REF nonlinear_elasticity.cc:410-499, 1153-1211]
  cfg4    BASELINE configuration 4: 59^3 Q2 block (5,055,477 DoFs), ONE Newton iteration of the same step: residual
          norm, right-hand side, Newton update (CG to 1e-10)               [REF nonlinear_elasticity.cc:444-487]
  cfg4s   BASELINE configuration 4, the first three Newmark steps of the bench's ramp, "Residual" = 1e-10
          (23 minutes per step on 8 cores)                                 [REF nonlinear_elasticity.cc:410-499]
  cfg2    BASELINE configuration 2: 40^3 Q1 cantilever of the linear model (206,763 DoFs), three theta-steps with the
          CG of the reference at an absolute tolerance of 1e-13            [REF linear_elasticity.cc:378-586]

Container only (minutes to an hour of CPU time on 8 cores; the mirror's dense matrices do not fit these sizes -- it
cross-checks the same code paths on the small meshes of make_golden.py).  Usage:

  python tests/golden/make_golden_big.py blk24 cfg2 cfg3 cfg4      # any subset; results are merged into the npz
  python tests/golden/make_golden_big.py thin                      # re-sample what is stored to the strides of STRIDES

Every case stores: the sampled node ids (every `stride`-th lattice line in each direction plus the last one), the
sampled values, the Euclidean norm of each whole vector and its inner product with a fixed deterministic weight vector
w_i = cos(0.37 i + 0.11) (pins the unsampled entries as well), and the Newton table."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

OUT = os.path.join(HERE, "baseline_sizes.npz")


def weights(n):
    return np.cos(0.37 * np.arange(n, dtype=np.float64) + 0.11)


def lattice_sample(npts, stride):
    """node ids of every `stride`-th lattice plane per direction (plus the last plane); npts = nodes per direction"""
    axes = []
    for m in npts:
        a = list(range(0, m, stride))
        if a[-1] != m - 1:
            a.append(m - 1)
        axes.append(np.array(a))
    if len(npts) == 2:
        ix, iy = np.meshgrid(axes[0], axes[1], indexing="ij")
        return np.sort((ix + npts[0] * iy).ravel()).astype(np.int32)
    ix, iy, iz = np.meshgrid(axes[0], axes[1], axes[2], indexing="ij")
    return np.sort((ix + npts[0] * (iy + npts[1] * iz)).ravel()).astype(np.int32)


def functionals(v):
    return np.array([np.linalg.norm(v), float(v @ weights(v.size))])


def distortion(cells, seed=77, amp=0.08):
    """vertex displacements of the distorted block: amp x cell size x standard normal, seeded (the GPU test builds the same)"""
    return amp / cells * np.random.default_rng(seed).standard_normal(((cells + 1) ** 3, 3))


def run_nonlinear(name, cells, steps, traction, tol_lin, stride, out, distorted=False, dim=3, degree=2):
    P = O.Problem(O.make_desc(dim=dim, degree=degree, reps=(cells,) * dim), distortion(cells) if distorted else None)
    npts = (degree * cells + 1,) * dim
    ids = lattice_sample(npts, stride)
    out[name + "_cells"], out[name + "_nodes"], out[name + "_tol_lin"] = cells, ids, tol_lin
    out[name + "_traction"] = np.array(traction)
    logs, us, fs = [], [], []
    for s in range(steps):
        P.set_interface_traction(traction[s])
        t0 = time.perf_counter()
        rc, info = P.newmark_step(O.SOLVER_CG_JACOBI, tol_lin=tol_lin, max_it_mult=2.0)
        assert rc == 0 and info.converged == 1, (rc, info.converged)
        print("%s step %d: %d Newton / %d CG iterations, %.0f s" % (name, s, info.newton_iterations, info.lin_its_total,
                                                                     time.perf_counter() - t0), flush=True)
        logs.append([info.newton_iterations, info.assemblies, info.res_norm, info.res_abs, info.upd_norm, info.upd_abs])
        u = P.vec(O.V_U).reshape(-1, dim)
        us.append(u[ids].copy())
        fs.append(np.stack([functionals(P.vec(w)) for w in (O.V_U, O.V_V, O.V_A)]))
    out[name + "_log"], out[name + "_u"], out[name + "_fun"] = np.array(logs), np.array(us), np.array(fs)
    out[name + "_v"] = P.vec(O.V_V).reshape(-1, dim)[ids].copy()
    out[name + "_a"] = P.vec(O.V_A).reshape(-1, dim)[ids].copy()


def run_one_newton_iteration(name, cells, traction, tol_lin, stride, out):
    P = O.Problem(O.make_desc(dim=3, degree=2, reps=(cells,) * 3))
    ids = lattice_sample((2 * cells + 1,) * 3, stride)
    P.set_interface_traction(traction)
    P.update_acceleration()
    t0 = time.perf_counter()
    P.assemble()
    print("%s assembly %.0f s" % (name, time.perf_counter() - t0), flush=True)
    out[name + "_cells"], out[name + "_nodes"], out[name + "_tol_lin"] = cells, ids, tol_lin
    out[name + "_traction"] = np.array(traction)
    out[name + "_res_norm"] = P.residual_norm()
    rhs = P.vec(O.V_RHS)
    out[name + "_rhs"], out[name + "_rhs_fun"] = rhs.reshape(-1, 3)[ids].copy(), functionals(rhs)
    # the tangent through fixed vectors: K w (sampled + functionals) pins the assembled operator at full size
    Kw = P.spmv(weights(P.n))
    out[name + "_Kw"], out[name + "_Kw_fun"] = Kw.reshape(-1, 3)[ids].copy(), functionals(Kw)
    P.vec(O.V_NEWTON)[:] = 0.0
    t0 = time.perf_counter()
    rc, its, res = P.solve_linear(O.SOLVER_CG_JACOBI, tol_lin=tol_lin, max_it_mult=2.0)
    assert rc == 0
    print("%s solve: %d CG iterations, %.0f s" % (name, its, time.perf_counter() - t0), flush=True)
    du = P.vec(O.V_NEWTON)
    free = ~P.constrained
    out[name + "_upd"], out[name + "_upd_fun"] = du.reshape(-1, 3)[ids].copy(), functionals(du)
    out[name + "_upd_norm_unconstrained"] = np.linalg.norm(du[free])  # get_error_update :564-576


def run_linear(name, cells, steps, traction, abs_tol, stride, out):
    d = O.make_desc(dim=3, degree=1, reps=(cells,) * 3, hi=(10.0, 1.0, 1.0), theta=0.5)
    L = O.LinearProblem(d)
    ids = lattice_sample((cells + 1,) * 3, stride)
    inodes = L.interface_nodes
    ds, vs, fs, its_all = [], [], [], []
    for s in range(steps):
        L.vec(O.L_STRESS)[:] = 0
        for c in range(3):
            L.vec(O.L_STRESS)[inodes * 3 + c] = traction[c]
        t0 = time.perf_counter()
        rc, its, res = L.step(O.SOLVER_CG_JACOBI, True, abs_tol=abs_tol)
        assert rc == 0
        print("%s step %d: %d CG iterations, %.0f s" % (name, s, its, time.perf_counter() - t0), flush=True)
        ds.append(L.vec(O.L_D).reshape(-1, 3)[ids].copy())
        vs.append(L.vec(O.L_V).reshape(-1, 3)[ids].copy())
        fs.append(np.stack([functionals(L.vec(O.L_D)), functionals(L.vec(O.L_V))]))
        its_all.append(its)
    out[name + "_cells"], out[name + "_nodes"], out[name + "_abs_tol"] = cells, ids, abs_tol
    out[name + "_traction"] = np.array(traction)
    out[name + "_d"], out[name + "_v"], out[name + "_fun"] = np.array(ds), np.array(vs), np.array(fs)


STRIDES = {"blk24": (49, 4), "blk24d": (49, 4), "q1_48": (49, 6), "q2_2d": (601, 20), "cfg3": (69, 8), "cfg4": (119, 14), "cfg4s": (119, 14), "cfg2": (41, 4)}  # nodes per direction, stride


def thin(out):
    """re-sample stored cases to the strides of STRIDES (a case generated with a finer stride keeps a subset)"""
    for name, (m, stride) in STRIDES.items():
        if name + "_nodes" not in out:
            continue
        dim = 2 if name + "_u" in out and np.shape(out[name + "_u"])[-1] == 2 else 3
        keep_ids = lattice_sample((m,) * dim, stride)
        ids = out[name + "_nodes"]
        sel = np.nonzero(np.isin(ids, keep_ids))[0]
        assert len(sel) == len(keep_ids), (name, len(sel), len(keep_ids))
        for k in list(out):
            if k.startswith(name + "_") and k != name + "_nodes" and np.ndim(out[k]) >= 2 and len(ids) in np.shape(out[k]):
                ax = list(np.shape(out[k])).index(len(ids))
                out[k] = np.take(out[k], sel, axis=ax)
        out[name + "_nodes"] = ids[sel]


def main():
    O.lib().orc_set_threads(len(os.sched_getaffinity(0)))
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for case in sys.argv[1:]:
        t0 = time.perf_counter()
        if case == "blk24":
            run_nonlinear("blk24", 24, 2, [(0.0, -2e3, 0.0)] * 2, 1e-12, STRIDES["blk24"][1], out)
        elif case == "blk24d":  # the same block with distorted cells: no cell is a box, the general-geometry kernels run
            run_nonlinear("blk24d", 24, 2, [(0.0, -2e3, 0.0)] * 2, 1e-12, STRIDES["blk24d"][1], out, distorted=True)
        elif case == "q1_48":  # other element families on the multigrid path (their smoother multiplies with the assembled matrix)
            run_nonlinear("q1_48", 48, 2, [(0.0, -2e3, 0.0)] * 2, 1e-12, STRIDES["q1_48"][1], out, degree=1)
        elif case == "q2_2d":
            run_nonlinear("q2_2d", 300, 2, [(0.0, -2e3)] * 2, 1e-12, STRIDES["q2_2d"][1], out, dim=2)
        elif case == "cfg3":
            run_nonlinear("cfg3", 34, 3, [(0.0, -2e2 * (k + 1), 0.0) for k in range(3)], 1e-10, STRIDES["cfg3"][1], out)
        elif case == "cfg4":
            run_one_newton_iteration("cfg4", 59, (0.0, -2e2, 0.0), 1e-10, STRIDES["cfg4"][1], out)
        elif case == "cfg4s":
            run_nonlinear("cfg4s", 59, 3, [(0.0, -2e2 * (k + 1), 0.0) for k in range(3)], 1e-10, STRIDES["cfg4s"][1], out)
        elif case == "cfg2":
            run_linear("cfg2", 40, 3, (0.0, -200.0, 0.0), 1e-13, STRIDES["cfg2"][1], out)
        elif case == "thin":
            thin(out)
        elif case == "tiny":  # seconds: exercises the script itself
            run_nonlinear("tiny", 4, 1, [(0.0, -2e3, 0.0)], 1e-12, 2, out)
            run_one_newton_iteration("tiny1", 4, (0.0, -2e2, 0.0), 1e-10, 2, out)
            for k in [k for k in out if k.startswith("tiny")]:
                del out[k]
        else:
            raise SystemExit("unknown case " + case)
        print("%s done in %.0f s" % (case, time.perf_counter() - t0), flush=True)
        np.savez_compressed(OUT, **out)
    print(json.dumps({k: list(np.shape(v)) for k, v in sorted(out.items())}))
    print("%s: %.0f kB" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
