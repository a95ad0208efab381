"""GPU tests of edge cases and of size-independent properties at a BASELINE configuration size (config 3: 34^3 Q2
cells, 985,527 DoFs), where the CPU oracle would take too long to be the checker:
symmetry and linearity of the assembled operator through the SpMV kernel, the solved system's true residual,
invariance of assembly and SpMV under the slab decomposition, bitwise repeatability.
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

M = load_pkg()
pytestmark = pytest.mark.gpu


def test_zero_load_step_follows_reference_escape_hatches():
    """no traction, zero state: residual 0 -> the Newton loop ends through the absolute escape hatches
    (nonlinear_elasticity.cc:459-463) after one (trivial) linear solve, for both preconditioners"""
    for precond in (0, 1):
        G = M.Context(dim=3, degree=2, reps=(3, 2, 2), hi=(0.3, 0.2, 0.2))
        G.set_tuning("precond", precond)
        rc, info = G.newmark_step()
        assert rc == 0 and info.converged == 1
        assert info.newton_iterations == 1 and info.assemblies == 2 and info.lin_its_total == 0
        assert info.res_abs == 0.0 and np.all(G.get(M.V_U) == 0.0)


def test_mesh_without_interface_and_single_cell():
    roles = [O.FACE_CLAMPED, 0, 0, 0, 0, 0]
    G = M.Context(dim=3, degree=1, reps=(2, 2, 2), face_role=roles, body_force=(0.0, -9.81, 0.0))
    P = O.Problem(O.make_desc(dim=3, degree=1, reps=(2, 2, 2), face_role=roles, body_force=(0.0, -9.81, 0.0)))
    ids, xyz = G.interface()
    assert len(ids) == 0 and xyz.shape == (0, 3)
    G.set_interface_traction(np.zeros((0, 3)))
    assert G.get_interface_displacement().shape == (0, 3)
    rc, info = G.newmark_step(tol_lin=1e-12)  # driven by the body force only
    rc_o, _ = P.newmark_step(O.SOLVER_DIRECT)
    assert rc == 0 and rc_o == 0
    assert np.abs(G.get(M.V_U) - P.vec(O.V_U)).max() / np.abs(P.vec(O.V_U)).max() < 1e-8
    # one cell, highest supported degrees
    for dim, p in ((2, 4), (3, 2)):
        G1 = M.Context(dim=dim, degree=p, reps=(1,) * dim, hi=(0.1,) * dim)
        P1 = O.Problem(O.make_desc(dim=dim, degree=p, reps=(1,) * dim, hi=(0.1,) * dim))
        t = (0.0, -500.0, 0.0)[:dim]
        G1.set_interface_traction(t)
        P1.set_interface_traction(t)
        assert G1.newmark_step(tol_lin=1e-12, max_it_mult=3.0)[0] == 0 and P1.newmark_step(O.SOLVER_DIRECT)[0] == 0
        assert np.abs(G1.get(M.V_U) - P1.vec(O.V_U)).max() / np.abs(P1.vec(O.V_U)).max() < 1e-8


@pytest.fixture(scope="module")
def config3():
    n = 34
    G = M.Context(dim=3, degree=2, reps=(n, n, n))
    assert G.n == 985527 and G.nnz == 183117753  # SURVEY.md section 8 size table
    rng = np.random.default_rng(1234)
    free = ~G.constrained
    G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * free)
    G.set(M.V_V_OLD, 0.1 * rng.standard_normal(G.n))
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.update_acceleration()
    G.assemble()
    return G


def test_fullsize_operator_symmetry_and_linearity(config3):
    G = config3
    rng = np.random.default_rng(4321)
    x, y = rng.standard_normal(G.n), rng.standard_normal(G.n)
    Kx, Ky = G.spmv(x), G.spmv(y)
    assert abs(y @ Kx - x @ Ky) / abs(y @ Kx) < 1e-11  # symmetric tangent (mirrored lower triangle, :1033-1035)
    Kz = G.spmv(2.5 * x - 0.75 * y)
    assert np.abs(Kz - (2.5 * Kx - 0.75 * Ky)).max() / np.abs(Kz).max() < 1e-12
    cons = G.constrained
    e = np.zeros(G.n)
    e[cons] = rng.standard_normal(cons.sum())
    Ke = G.spmv(e)
    assert np.all(Ke[~cons] == 0.0) and np.all(Ke[cons] * e[cons] > 0)  # constrained rows: positive diagonal only
    # block-CSR cross-check kernel on the same matrix
    G.set_tuning("spmv_variant", 1)
    assert np.abs(G.spmv(x) - Kx).max() / np.abs(Kx).max() < 1e-13
    G.set_tuning("spmv_variant", 3)


def test_fullsize_solve_true_residual_and_preconditioners_agree(config3):
    G = config3
    b = G.get(M.V_RHS)
    sols = {}
    for precond in (1, 0):
        G.set_tuning("precond", precond)
        G.set(M.V_NEWTON, np.zeros(G.n))
        rc, its, res = G.cg_solve(rel_tol=1e-9)
        assert rc == 0
        x = G.get(M.V_NEWTON)
        true_res = np.linalg.norm(b - G.spmv(x))
        assert true_res <= 1.5e-9 * np.linalg.norm(b)  # recursion residual == true residual
        assert abs(true_res - res) / res < 1e-3
        sols[precond] = (x, its)
    assert np.abs(sols[1][0] - sols[0][0]).max() / np.abs(sols[0][0]).max() < 1e-6
    assert sols[1][1] * 8 < sols[0][1]  # multigrid: mesh-independent iteration count
    G.set_tuning("precond", 1)


def test_fullsize_decomposition_invariance(config3):
    """4 slabs reproduce the single-slab residual vector and operator action at 1M DoFs"""
    G = config3
    n = 34
    G4 = M.Context(dim=3, degree=2, reps=(n, n, n), slabs=4)
    for k in (M.V_U, M.V_V_OLD):
        G4.set(k, G.get(k))
    G4.set_interface_traction((0.0, -2e3, 0.0))
    G4.update_acceleration()
    rn4 = G4.assemble()
    r1, r4 = G.get(M.V_RHS), G4.get(M.V_RHS)
    assert np.abs(r4 - r1).max() / np.abs(r1).max() < 1e-12
    assert abs(rn4 - np.linalg.norm(r1[~G.constrained])) / rn4 < 1e-12
    x = np.random.default_rng(5).standard_normal(G.n)
    assert np.abs(G4.spmv(x) - G.spmv(x)).max() / np.abs(G.spmv(x)).max() < 1e-13
    # (the slabs are below the size at which the smoother multiplies matrix-free: they smooth with the assembled operator, the
    # undecomposed run with the 27-point form of it -- on this grid-scale-noisy state the two differ by a percent, so the Krylov
    # paths differ: both solutions are held against the TRUE residual bound of the stopping rule, and against each other at
    # the accuracy that bound implies)
    rc, its4, _ = G4.cg_solve(rel_tol=1e-10)
    G.set(M.V_NEWTON, np.zeros(G.n))
    rc1, its1, _ = G.cg_solve(rel_tol=1e-10)
    assert rc == 0 and rc1 == 0 and abs(its4 - its1) <= 2
    b = G.get(M.V_RHS)
    for H in (G, G4):
        assert np.linalg.norm(b - H.spmv(H.get(M.V_NEWTON))) <= 1.5e-10 * np.linalg.norm(b)
    assert np.abs(G4.get(M.V_NEWTON) - G.get(M.V_NEWTON)).max() / np.abs(G.get(M.V_NEWTON)).max() < 1e-6


def test_fp32_stored_smoother_matrix_is_only_a_preconditioner_change(config3):
    """opt-in: the V-cycle's smoother multiplies with an fp32-rounded copy of the level matrices.  The CG operator,
    its residual and the stopping rule stay fp64, so the solution satisfies the same true-residual bound and the
    iteration count moves by at most a few"""
    G = config3
    b = G.get(M.V_RHS)
    out = {}
    for bits in (64, 32):
        G.set_tuning("precond_storage", bits)
        G.set(M.V_NEWTON, np.zeros(G.n))
        rc, its, res = G.cg_solve(rel_tol=1e-10)
        assert rc == 0
        x = G.get(M.V_NEWTON)
        assert np.linalg.norm(b - G.spmv(x)) <= 1.5e-10 * np.linalg.norm(b)  # fp64 product in mi_spmv
        out[bits] = (x, its)
    G.set_tuning("precond_storage", 64)
    assert abs(out[32][1] - out[64][1]) <= 2
    assert np.abs(out[32][0] - out[64][0]).max() / np.abs(out[64][0]).max() < 1e-7


def test_headline_size_solver_policies_do_not_change_the_solution():
    """BASELINE configuration 4 (59^3 Q2 cells) under the policies bench.py and the executable run with -- start vectors
    from the previous step, coarse operators kept over the steps -- against the library's plain setting (zero start,
    operators rebuilt every step): four ramp steps at a tight linear tolerance, same Newton tables, states equal to 1e-8"""
    n, out = 59, {}
    for name, start, every in (("plain", 0, 1), ("policies", 2, 8)):
        G = M.Context(dim=3, degree=2, reps=(n, n, n))
        G.set_tuning("cg_warm_start", start)
        G.set_tuning("mg_refresh_every", every)
        G.reset_timings()
        table, its = [], 0
        for k in range(4):
            G.set_interface_traction((0.0, -2e3 * (k + 1) / 10.0, 0.0))
            rc, info = G.newmark_step(tol_lin=1e-10, max_it_mult=1.0)
            assert rc == 0 and info.converged == 1
            table.append((info.newton_iterations, info.assemblies))
            its += info.lin_its_total
        out[name] = (table, its, G.get(M.V_U), G.get(M.V_V), G.get_tuning("count_mg_refresh"))
        G.close()
    assert out["policies"][0] == out["plain"][0]
    assert out["plain"][4] == 4 and out["policies"][4] == 1
    assert out["policies"][1] <= out["plain"][1]
    for k in (2, 3):
        assert np.abs(out["policies"][k] - out["plain"][k]).max() / np.abs(out["plain"][k]).max() < 1e-8


def test_headline_size_invariants():
    """BASELINE configuration 4 (59^3 Q2 cells, the size bench.py runs): sizes of SURVEY.md section 8, bitwise
    repeatable assembly, symmetric operator, a full Newmark step with the reference's Newton bookkeeping and a
    true-residual check of its last linear solve, and agreement of a 2-slab decomposition"""
    n = 59
    G = M.Context(dim=3, degree=2, reps=(n, n, n))
    assert (G.n, G.nnz, G.ncells) == (5055477, 952414353, 205379)
    G.set_interface_traction((0.0, -2e3, 0.0))
    rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0)
    assert rc == 0 and info.converged == 1
    assert info.assemblies == info.newton_iterations + 1  # nonlinear_elasticity.cc:446-469
    assert 10 <= info.lin_its_total <= 80  # mesh-independent multigrid-PCG (a Jacobi-PCG needs > 1000 here)
    u = G.get(M.V_U)
    assert np.all(np.isfinite(u)) and np.all(u[G.constrained] == 0.0) and np.abs(u).max() > 0
    # the state after the step: assemble twice -> identical bits (colouring, fixed reduction order)
    G.update_acceleration()
    rn1 = G.assemble()
    r1 = G.get(M.V_RHS)
    rn2 = G.assemble()
    assert rn1 == rn2 and np.array_equal(r1, G.get(M.V_RHS))
    rng = np.random.default_rng(99)
    x, y = rng.standard_normal(G.n), rng.standard_normal(G.n)
    Kx, Ky = G.spmv(x), G.spmv(y)
    assert abs(y @ Kx - x @ Ky) / abs(y @ Kx) < 1e-11
    # the smoother's operator (matrix-free from the quadrature-point records, active at this size) is the assembled matrix
    assert G.get_tuning("smoother_operator_active") == 2
    G.set_tuning("spmv_variant", 4)
    assert np.abs(G.spmv(x) - Kx).max() / np.abs(Kx).max() < 1e-13
    G.set_tuning("spmv_variant", 3)
    G.set(M.V_NEWTON, np.zeros(G.n))
    rc, its, res = G.cg_solve(rel_tol=1e-8)
    assert rc == 0
    true_res = np.linalg.norm(r1 - G.spmv(G.get(M.V_NEWTON)))
    assert true_res <= 1.5e-8 * np.linalg.norm(r1) and abs(true_res - res) / res < 1e-3
    # the reference's start vector for the later solves of the step: same Newton bookkeeping, more linear iterations
    Gw = M.Context(dim=3, degree=2, reps=(n, n, n))
    Gw.set_tuning("cg_warm_start", 1)
    Gw.set_interface_traction((0.0, -2e3, 0.0))
    rcw, infow = Gw.newmark_step(tol_lin=1e-6, max_it_mult=1.0)
    assert rcw == 0 and infow.newton_iterations == info.newton_iterations and infow.lin_its_total > info.lin_its_total
    assert np.abs(Gw.get(M.V_U) - u).max() / np.abs(u).max() < 1e-5  # Residual = 1e-6 regime
    del Gw
    # two slabs: same residual vector and the same Newton/CG bookkeeping for the same step
    G2 = M.Context(dim=3, degree=2, reps=(n, n, n), slabs=2)
    G2.set_interface_traction((0.0, -2e3, 0.0))
    rc2, info2 = G2.newmark_step(tol_lin=1e-6, max_it_mult=1.0)
    assert rc2 == 0 and info2.newton_iterations == info.newton_iterations
    assert abs(info2.lin_its_total - info.lin_its_total) <= 2
    assert np.abs(G2.get(M.V_U) - u).max() / np.abs(u).max() < 1e-5  # Residual = 1e-6 regime


def test_contexts_release_their_device_memory():
    """create / step / destroy many contexts (single, emulated slabs, multigrid levels, linear model, fp32 smoother
    copy, snapshots): free device memory returns to where it started"""
    import ctypes as C
    import torch
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]

    def cycle(k):
        G = M.Context(dim=3, degree=2, reps=(6, 5, 6), slabs=1 + k % 3)
        G.set_interface_traction((0.0, -1e3, 0.0))
        if k % 4 == 1:
            G.set_tuning("precond_storage", 32)
        rc, info = G.newmark_step(tol_lin=1e-8)
        assert rc == 0 and info.converged == 1
        G.state_save()
        if k % 2:
            assert L.mi_linear_setup(G.h, 0.5) == 0
        G.close()

    cycle(0)  # first use pays one-off allocations of the runtime (code objects, RCCL-free pools)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(24):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "leaked %.1f MB over 24 contexts" % ((free0 - free1) / 2 ** 20)


def test_config2_linear_model_at_full_size():
    """BASELINE configuration 2: linear_elasticity 3D Q1 cantilever, 40^3 cells = 206,763 DoFs (15,944,049 nnz), prescribed
    constant traction.  Size-independent properties of the device path (the oracle's direct solver is out of reach
    here): symmetry of K and M, rigid translations in the kernel of K, total mass, the solved theta-steps' true residual
    against the exported stepping matrix, the displacement update, and energy balance of the theta = 1/2 scheme."""
    import ctypes as C
    import scipy.sparse as sp
    n, dt, theta, rho = 40, 0.005, 0.5, 1000.0
    roles = [O.FACE_CLAMPED] + [O.FACE_INTERFACE] * 5
    G = M.Context(dim=3, degree=1, reps=(n, n, n), lo=(0, 0, 0), hi=(10.0, 1.0, 1.0), face_role=roles, rho=rho, delta_t=dt)
    assert G.n == 206763 and G.nnz == 15944049  # SURVEY.md section 8 size table
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mi_linear_matrix_get_csr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double)]
    assert L.mi_linear_setup(G.h, theta) == 0, L.mi_last_error(G.h)

    def mat(which):
        rp, col, val = np.zeros(G.n + 1, dtype=np.int64), np.zeros(G.nnz, dtype=np.int32), np.zeros(G.nnz)
        assert L.mi_linear_matrix_get_csr(G.h, which, rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                          col.ctypes.data_as(C.POINTER(C.c_int32)), M._dp(val)) == 0
        return sp.csr_matrix((val, col, rp), shape=(G.n, G.n))

    K, Mm, S = mat(0), mat(1), mat(2)
    for A in (K, Mm, S):
        assert abs(A - A.T).max() / abs(A).max() < 1e-13
    for c in range(3):  # a rigid translation carries no strain energy; the mass of every component is rho * volume
        e = np.zeros(G.n)
        e[c::3] = 1.0
        assert np.abs(K @ e).max() / abs(K).max() < 1e-11
        assert abs(e @ (Mm @ e) / (rho * 10.0) - 1.0) < 1e-12
    con = G.constrained
    # boundary values keep the diagonal (the device sums M_e + theta^2 dt^2 K_e cell by cell: equal up to rounding)
    assert np.allclose(S.diagonal()[con], (Mm + (theta * dt) ** 2 * K).diagonal()[con], rtol=1e-13, atol=0.0)
    G.set_interface_traction((0.0, -200.0, 0.0))
    d_old, v_old = np.zeros(G.n), np.zeros(G.n)
    energy = [0.0]
    for step in range(3):
        its, res = C.c_int(0), C.c_double(0)
        assert L.mi_linear_step(G.h, 1, 1e-10, G.n, C.byref(its), C.byref(res)) == 0, L.mi_last_error(G.h)
        d, v, rhs, f = G.get(0), G.get(2), G.get(9), G.get(4)  # displacement, velocity, system_rhs, F_n+1 (old_stress)
        assert its.value > 0 and res.value <= 1e-10
        assert np.linalg.norm(S @ v - rhs) <= 2e-10 and np.all(v[con] == 0)       # the reference's absolute 1e-10 (:542)
        assert np.abs(d - (d_old + dt * (theta * v + (1 - theta) * v_old))).max() <= 1e-15 * max(1.0, np.abs(d).max())
        # theta = 1/2 conserves energy: E_n+1 - E_n = work of the mean load along the displacement increment
        f_prev = f if step > 0 else np.zeros(G.n)  # F_0 = 0 (:243)
        E = 0.5 * v @ (Mm @ v) + 0.5 * d @ (K @ d)
        work = 0.5 * (f + f_prev) @ (d - d_old)
        assert abs((E - energy[-1]) - work) <= 1e-6 * max(abs(work), 1e-30), (step, E - energy[-1], work)
        energy.append(E)
        d_old, v_old = d, v
    assert energy[-1] > 0
