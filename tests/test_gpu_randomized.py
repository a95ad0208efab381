"""Seeded random sweep of the assembly + operator parity (GPU through the C-ABI vs the CPU oracle): random
dimension, degree, cell counts, box, vertex distortion, boundary roles, material, Newmark parameters, body force,
state vectors, tractions and slab count.  Broad coverage of parameter combinations the hand-picked cases miss;
every case is reproducible from its seed.  Tolerances: residual vector and operator action 1e-11 relative (fp64,
different summation orders)."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

M = load_pkg()
pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.integers(2, 4))
    p = int(rng.integers(1, 5 if dim == 2 else 3))
    hi_cells = 5 if (dim == 2 or p == 1) else 3
    reps = tuple(int(rng.integers(1, hi_cells + 1)) for _ in range(dim))
    lo = tuple(float(x) for x in rng.uniform(-1.0, 1.0, dim))
    h = float(rng.uniform(0.05, 0.3))
    hi = tuple(lo[d] + h * reps[d] * float(rng.uniform(0.7, 1.4)) for d in range(dim))
    choices = [0, O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE]
    roles = [int(rng.choice(choices)) for _ in range(2 * dim)] + [0] * (6 - 2 * dim)
    if dim == 3 and rng.random() < 0.5:
        roles[4] = roles[5] = O.FACE_ZCLAMP
    roles[int(rng.integers(0, 2 * (dim - 1)))] = O.FACE_CLAMPED  # at least one clamped side
    kw = dict(mu=float(10 ** rng.uniform(4, 7)), nu=float(rng.uniform(0.05, 0.45)), rho=float(rng.uniform(0, 3000)),
              body_force=tuple(float(x) for x in rng.uniform(-10, 10, 3)), beta=float(rng.uniform(0.25, 0.5)),
              gamma=float(rng.uniform(0.5, 0.9)), delta_t=float(10 ** rng.uniform(-4, -1)))
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = 0.12 * h * rng.uniform(-1, 1, (nverts, dim))
    slabs = int(rng.integers(1, min(3, reps[-1]) + 1))
    return rng, dim, p, reps, lo, hi, roles, kw, perturb, slabs, h


@pytest.mark.parametrize("seed", range(48))
def test_random_configuration(seed):
    rng, dim, p, reps, lo, hi, roles, kw, perturb, slabs, h = _case(seed)
    P = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, **kw), perturb)
    G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, perturb=perturb, slabs=slabs, **kw)
    assert (G.n, G.nnz) == (P.n, P.nnz)
    assert np.array_equal(G.constrained, P.constrained)
    ids, _ = G.interface()
    assert np.array_equal(ids, P.interface_nodes)
    free = ~P.constrained
    hp = h / p
    for k, v in {O.V_U: 0.02 * hp * rng.standard_normal(P.n) * free, O.V_DELTA: 0.01 * hp * rng.standard_normal(P.n) * free,
                 O.V_V_OLD: rng.standard_normal(P.n), O.V_A_OLD: 10 * rng.standard_normal(P.n)}.items():
        P.vec(k)[:] = v
        G.set(k, v)
    t = kw["mu"] * 1e-3 * rng.standard_normal((len(ids), dim))
    P.set_interface_traction(t)
    G.set_interface_traction(t)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    rn = G.assemble()
    r_o = P.vec(O.V_RHS)
    scale = max(np.abs(r_o).max(), 1e-300)
    assert np.abs(G.get(M.V_RHS) - r_o).max() / scale < 1e-11, "seed %d: dim %d p %d reps %s slabs %d" % (seed, dim, p, reps, slabs)
    assert abs(rn - P.residual_norm()) <= 1e-11 * max(P.residual_norm(), scale)
    x = rng.standard_normal(P.n)
    y_o = P.csr() @ x
    assert np.abs(G.spmv(x) - y_o).max() / np.abs(y_o).max() < 1e-11


@pytest.mark.parametrize("seed", range(0, 48, 3))
def test_random_configuration_newmark_step(seed):
    """one full Newmark step (Newton + CG + updates) of the same random configurations against the oracle with its
    direct solver; linear tolerance 1e-12 on the device, displacement agreement 1e-7 relative"""
    rng, dim, p, reps, lo, hi, roles, kw, perturb, slabs, h = _case(seed)
    P = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, **kw), perturb)
    G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, perturb=perturb, slabs=slabs, **kw)
    G.set_tuning("precond", (seed // 3) % 2)  # small meshes default to Jacobi: keep the V-cycle covered as well
    ids, _ = G.interface()
    t = kw["mu"] * 2e-4 * rng.standard_normal((len(ids), dim))
    P.set_interface_traction(t)
    G.set_interface_traction(t)
    rc_o, info_o = P.newmark_step(O.SOLVER_DIRECT if P.n < 3000 else O.SOLVER_CG_SSOR, tol_lin=1e-13, max_it_mult=4.0)
    rc, info = G.newmark_step(tol_lin=1e-12, max_it_mult=4.0)
    assert rc_o == 0 and rc == 0 and info.converged == 1
    assert info.newton_iterations == info_o.newton_iterations
    for k in (M.V_U, M.V_V, M.V_A):
        ref = P.vec(k)
        assert np.abs(G.get(k) - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-300), (seed, k)


def _q2_3d_seeds(count, start=300):
    """seeds whose random configuration is a 3D Q2 mesh (where the matrix-free fine level exists)"""
    out, seed = [], start
    while len(out) < count:
        _, dim, p, *_ = _case(seed)
        if dim == 3 and p == 2:
            out.append(seed)
        seed += 1
    return out


@pytest.mark.parametrize("seed", _q2_3d_seeds(12))
def test_random_configuration_matrix_free_fine_level(seed):
    """round 6: the same sweep over random 3D Q2 configurations (boxes of random shape, distorted cells, random boundary
    roles incl. the z-clamp, random material / Newmark parameters / body force, 1-3 slabs) with the fine level matrix-free
    (tuning "fine_level" 1): residual and operator action against the oracle's assembled system to 1e-11, the nodes' diagonal
    blocks (mf_diag) to 1e-11, then one Newmark step against the oracle's (multigrid or Jacobi by the seed) to 1e-7"""
    rng, dim, p, reps, lo, hi, roles, kw, perturb, slabs, h = _case(seed)
    P = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, **kw), perturb)
    G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, perturb=perturb, slabs=slabs, **kw)
    G.set_tuning("precond", seed % 2)
    G.set_tuning("fine_level", 1)
    G.set_tuning("mf_diag_lag", (seed // 2) % 2)
    ids, _ = G.interface()
    free = ~P.constrained
    hp = h / p
    state = {O.V_U: 0.02 * hp * rng.standard_normal(P.n) * free, O.V_DELTA: 0.01 * hp * rng.standard_normal(P.n) * free,
             O.V_V_OLD: rng.standard_normal(P.n), O.V_A_OLD: 10 * rng.standard_normal(P.n)}
    for k, v in state.items():
        P.vec(k)[:] = v
        G.set(k, v)
    t = kw["mu"] * 1e-3 * rng.standard_normal((len(ids), dim))
    P.set_interface_traction(t)
    G.set_interface_traction(t)
    P.update_acceleration()
    P.assemble()
    G.newton_begin_step()  # ("mf_diag_lag": a new step forms the diagonal blocks)
    for k in (O.V_DELTA,):
        G.set(k, state[k])
    G.update_acceleration()
    rn = G.assemble()
    r_o = P.vec(O.V_RHS)
    scale = max(np.abs(r_o).max(), 1e-300)
    tag = "seed %d: reps %s slabs %d" % (seed, reps, slabs)
    assert np.abs(G.get(M.V_RHS) - r_o).max() / scale < 1e-11, tag
    assert abs(rn - P.residual_norm()) <= 1e-11 * max(P.residual_norm(), scale)
    K = P.csr()
    x = rng.standard_normal(P.n)
    y_o = K @ x
    assert np.abs(G.spmv(x) - y_o).max() / np.abs(y_o).max() < 1e-11, tag
    nn, base = P.n // 3, np.arange(P.n // 3) * 3
    D_o = np.stack([np.stack([np.asarray(K[base + i, base + j]).ravel() for j in range(3)], -1) for i in range(3)], -2)
    assert np.abs(G.diagonal_blocks() - D_o).max() / np.abs(D_o).max() < 1e-11, tag
    # a whole step from rest
    P2 = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, **kw), perturb)
    for k in (M.V_U, M.V_DELTA, M.V_V_OLD, M.V_A_OLD, M.V_U_OLD, M.V_V, M.V_A, M.V_NEWTON):
        G.set(k, np.zeros(G.n))
    t = kw["mu"] * 2e-4 * rng.standard_normal((len(ids), dim))
    P2.set_interface_traction(t)
    G.set_interface_traction(t)
    rc_o, info_o = P2.newmark_step(O.SOLVER_DIRECT if P2.n < 3000 else O.SOLVER_CG_SSOR, tol_lin=1e-13, max_it_mult=4.0)
    rc, info = G.newmark_step(tol_lin=1e-12, max_it_mult=4.0)
    assert rc_o == 0 and rc == 0 and info.converged == 1 and info.newton_iterations == info_o.newton_iterations, tag
    for k in (M.V_U, M.V_V, M.V_A):
        ref = P2.vec(k)
        assert np.abs(G.get(k) - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-300), (tag, k)


@pytest.mark.parametrize("seed", range(100, 116))
def test_random_linear_model(seed):
    """the linear theta-model on random configurations (incl. slabs): 3 steps with random coupling data, alternating
    the 'Stress' (consistent) and 'Force' paths, against the oracle's direct solve"""
    import ctypes as C
    rng, dim, p, reps, lo, hi, roles, kw, perturb, slabs, h = _case(seed)
    theta = float(rng.uniform(0.5, 1.0))
    lin_kw = dict(mu=kw["mu"], nu=kw["nu"], rho=max(kw["rho"], 1.0), body_force=kw["body_force"], delta_t=kw["delta_t"])
    P = O.LinearProblem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, theta=theta, **lin_kw))
    G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, slabs=slabs, **lin_kw)
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    assert L.mi_linear_setup(G.h, theta) == 0, L.mi_last_error(G.h)
    ids = P.interface_nodes
    for step in range(3):
        consistent = step != 1
        t = kw["mu"] * 1e-4 * rng.standard_normal((len(ids), dim))
        P.vec(O.L_STRESS)[:] = 0
        for c in range(dim):
            P.vec(O.L_STRESS)[ids * dim + c] = t[:, c]
        G.set_interface_traction(t)
        assert P.step(O.SOLVER_DIRECT, consistent)[0] == 0
        its, res = C.c_int(0), C.c_double(0)
        scale = max(np.abs(P.vec(O.L_RHS)).max(), 1e-30)
        rc = L.mi_linear_step(G.h, int(consistent), 1e-13 * scale, G.n * 10, C.byref(its), C.byref(res))
        assert rc == 0, L.mi_last_error(G.h)
        for vo, vg in ((O.L_D, 0), (O.L_V, 2)):
            ref = P.vec(vo)
            assert np.abs(G.get(vg) - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-300), (seed, step, vo)


@pytest.mark.parametrize("seed", range(200, 214))
def test_random_multigrid_solve(seed):
    """multigrid-PCG on random mid-size configurations (anisotropic cell counts and sizes, distorted cells, random
    Dirichlet sides, slabs): converges without the Jacobi fallback, to the same solution as Jacobi-PCG, in a bounded
    number of iterations that does not depend on the number of slabs"""
    rng = np.random.default_rng(seed)
    dim = 3 if seed % 3 else 2
    p = int(rng.integers(1, 3)) if dim == 3 else int(rng.integers(1, 4))
    reps = tuple(int(rng.integers(4, 11 if dim == 3 else 25)) for _ in range(dim))
    hcell = rng.uniform(0.02, 0.1, dim)
    hi = tuple(float(hcell[d] * reps[d]) for d in range(dim))
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, int(rng.choice([0, O.FACE_CLAMPED, O.FACE_INTERFACE])), O.FACE_INTERFACE,
             int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE])), int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE]))]
    kw = dict(mu=float(10 ** rng.uniform(5, 7)), nu=float(rng.uniform(0.2, 0.45)), rho=float(rng.uniform(100, 3000)),
              delta_t=float(10 ** rng.uniform(-3, -1)))
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = 0.1 * hcell.min() * rng.uniform(-1, 1, (nverts, dim))
    slabs = int(rng.integers(2, 5))
    sols, its = {}, {}
    for tag, s, precond in (("mg1", 1, 1), ("mgN", slabs, 1), ("jac", 1, 0)):
        G = M.Context(dim=dim, degree=p, reps=reps, hi=hi, face_role=roles, perturb=perturb, slabs=s, **kw)
        G.set_tuning("precond", precond)
        r2 = np.random.default_rng(seed + 1)
        G.set(M.V_U, 0.01 * hcell.min() / p * r2.standard_normal(G.n) * ~G.constrained)
        ids, _ = G.interface()
        G.set_interface_traction(kw["mu"] * 1e-3 * r2.standard_normal((len(ids), dim)))
        G.update_acceleration()
        G.assemble()
        rc, n_it, res = G.cg_solve(rel_tol=1e-9, max_it=20000 if precond == 0 else 150)
        assert rc == 0, (seed, tag)
        sols[tag], its[tag] = G.get(M.V_NEWTON), n_it
    scale = np.abs(sols["jac"]).max()
    assert np.abs(sols["mg1"] - sols["jac"]).max() / scale < 1e-6
    assert np.abs(sols["mgN"] - sols["jac"]).max() / scale < 1e-6
    assert abs(its["mgN"] - its["mg1"]) <= 2, (seed, its)
    assert its["mg1"] <= 60, (seed, its, dim, p, reps)
