"""GPU parity tests: HIP hot path (through the C-ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (fp64 path):
  * element / global tangent and residual:   1e-12 relative (max-norm against the matrix/vector max)
  * SpMV:                                     1e-13 relative
  * converged linear / Newton solutions:      1e-8 relative with the linear tolerance tightened to 1e-12
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

M = load_pkg()
pytestmark = pytest.mark.gpu

TOL_ASM = 1e-12
TOL_SOL = 1e-8


def _pair(dim, degree, reps, perturb_amp=0.0, seed=0, roles=None, **kw):
    """oracle problem + device context on the same mesh"""
    lo = (0.0,) * dim
    hi = tuple(0.1 * r for r in reps)
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = None
    if perturb_amp:
        perturb = perturb_amp * 0.1 * np.random.default_rng(seed).standard_normal((nverts, dim))
    if roles is None:
        roles = [O.FACE_CLAMPED] + [O.FACE_INTERFACE] * 5
    gkw = {k: kw.pop(k) for k in ("slabs", "cut_axis") if k in kw}  # the device side alone: emulated slabs
    d = O.make_desc(dim=dim, degree=degree, reps=reps, lo=lo, hi=hi, face_role=roles, **kw)
    P = O.Problem(d, perturb)
    G = M.Context(dim=dim, degree=degree, reps=reps, lo=lo, hi=hi, face_role=roles, perturb=perturb, **kw, **gkw)
    return P, G


def _randomise_state(P, G, seed, amp_u=0.01):
    rng = np.random.default_rng(seed)
    n = P.n
    h = 0.1 / P.desc.degree
    free = ~P.constrained
    state = {
        O.V_U: amp_u * h * rng.standard_normal(n) * free,
        O.V_DELTA: 0.5 * amp_u * h * rng.standard_normal(n) * free,
        O.V_V_OLD: 0.1 * rng.standard_normal(n),
        O.V_A_OLD: rng.standard_normal(n),
    }
    for k, v in state.items():
        P.vec(k)[:] = v
        G.set(k, v)
    nif = len(P.interface_nodes)
    t = 2e3 * rng.standard_normal((nif, P.dim))
    P.set_interface_traction(t)
    G.set_interface_traction(t)


def _relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_mesh_numbering_matches_oracle():
    P, G = _pair(3, 2, (2, 3, 2), perturb_amp=0.1)
    assert (G.n, G.nnz, G.ncells) == (P.n, P.nnz, P.ncells)
    assert np.allclose(G.coords, P.coords, rtol=0, atol=1e-15)
    assert np.array_equal(G.constrained, P.constrained)
    ids, xyz = G.interface()
    assert np.array_equal(ids, P.interface_nodes) and np.allclose(xyz, P.coords[ids], atol=1e-15)


@pytest.mark.parametrize("dim,p", [(2, 1), (2, 2), (2, 3), (2, 4), (3, 1), (3, 2), (3, 3), (3, 4)])
def test_single_cell_tangent_and_residual(dim, p):
    """one distorted cell, no constraints: K_e and r_e against the as-written reference loop"""
    P, G = _pair(dim, p, (1,) * dim, perturb_amp=0.08, seed=10 + p, roles=[O.FACE_INTERFACE] * 6,
                 body_force=(3.0, -9.81, 1.5 if dim == 3 else 0.0))
    _randomise_state(P, G, seed=20 + p)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    rn = G.assemble()
    K_o, K_g = P.csr().toarray(), G.csr().toarray()
    assert _relmax(K_g, K_o) < TOL_ASM
    assert _relmax(G.get(M.V_RHS), P.vec(O.V_RHS)) < TOL_ASM
    assert abs(rn - P.residual_norm()) / P.residual_norm() < 1e-12
    assert _relmax(G.get(M.V_A), P.vec(O.V_A)) < 1e-15


@pytest.mark.parametrize("dim,p,reps", [(2, 1, (5, 4)), (2, 2, (4, 3)), (2, 3, (3, 3)), (2, 4, (2, 3)),
                                        (3, 1, (3, 4, 2)), (3, 2, (3, 2, 3)), (3, 2, (1, 1, 5)), (3, 3, (2, 3, 2)),
                                        (3, 4, (2, 1, 2))])
def test_global_assembly_with_constraints_and_traction(dim, p, reps):
    """all colours, Dirichlet rows (clamped + z-clamp), Neumann faces with the pull-back quirk"""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    P, G = _pair(dim, p, reps, perturb_amp=0.05, seed=3, roles=roles)
    _randomise_state(P, G, seed=4)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    rn = G.assemble()
    K_o, K_g = P.csr(), G.csr()
    assert np.array_equal(K_o.indptr, K_g.indptr) and np.array_equal(K_o.indices, K_g.indices)
    assert _relmax(K_g.data, K_o.data) < TOL_ASM
    r_o, r_g = P.vec(O.V_RHS), G.get(M.V_RHS)
    assert _relmax(r_g, r_o) < TOL_ASM
    assert np.all(r_g[P.constrained] == 0)
    assert abs(rn - P.residual_norm()) / P.residual_norm() < 1e-12
    # a second assembly reproduces the first bit for bit (colouring => deterministic summation order)
    G.assemble()
    assert np.array_equal(G.csr().data, K_g.data) and np.array_equal(G.get(M.V_RHS), r_g)


@pytest.mark.parametrize("dim,p,reps", [(2, 3, (4, 3)), (3, 1, (3, 3, 2)), (3, 2, (3, 2, 3)), (3, 3, (2, 2, 2))])
def test_residual_only_assembly_is_bitwise_the_full_residual(dim, p, reps):
    """mi_assemble_residual (the pass behind the Newton convergence check): same system_rhs and norm, bit for bit,
    and the tangent, its SpMV copy and the Jacobi diagonal keep the state of the last full assembly"""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    P, G = _pair(dim, p, reps, perturb_amp=0.05, seed=3, roles=roles)
    _randomise_state(P, G, seed=4)
    G.update_acceleration()
    rn_full = G.assemble()
    r_full, K_full = G.get(M.V_RHS), G.csr().data.copy()
    x = np.random.default_rng(1).standard_normal(G.n)
    y_full = G.spmv(x)
    # move the state, residual-only pass: rhs follows the new state, the tangent stays
    _randomise_state(P, G, seed=5)
    G.update_acceleration()
    rn_res = G.assemble_residual()
    r_res = G.get(M.V_RHS)
    assert np.array_equal(G.csr().data, K_full) and np.array_equal(G.spmv(x), y_full)
    rn_new = G.assemble()
    assert rn_res == rn_new and np.array_equal(r_res, G.get(M.V_RHS))
    assert not np.array_equal(r_res, r_full) and rn_res != rn_full
    assert not np.array_equal(G.csr().data, K_full)


def test_neumann_only_residual_isolated():
    """zero displacement and acceleration: rhs is exactly the interface load, total force = traction * area"""
    P, G = _pair(3, 2, (2, 2, 2))
    t = np.array([100.0, -2e3, 50.0])
    P.set_interface_traction(t)
    G.set_interface_traction(t)
    P.assemble()
    G.assemble()
    r = G.get(M.V_RHS).reshape(-1, 3)
    assert _relmax(r, P.vec(O.V_RHS).reshape(-1, 3)) < TOL_ASM
    # interface = 5 faces of the 0.2^3 cube; clamped rows carry no load, so compare with the oracle only,
    # and check the free-node sum against (area - share of clamped edge nodes) via the oracle value
    assert np.allclose(r.sum(0), P.vec(O.V_RHS).reshape(-1, 3).sum(0), rtol=1e-12)


@pytest.mark.parametrize("dim,p,reps,slabs,cut_axis", [(3, 2, (3, 2, 3), 1, 0), (2, 3, (4, 3), 1, 0), (3, 1, (3, 3, 4), 2, 0),
                                                       (2, 2, (6, 3), 3, 1), (3, 3, (2, 2, 2), 1, 0), (3, 2, (4, 2, 2), 2, 1)])
def test_face_pull_back_with_the_deformation_gradient_on_the_face(dim, p, reps, slabs, cut_axis):
    """the reference pulls the traction back with F of CELL quadrature point fq (nonlinear_elasticity.cc:
    825-827 against :902-903) -- reproduced by default; mi_set_tuning("correct_face_F", 1) evaluates F at the face
    quadrature point itself, which is what the oracle's correct_face_F switch does.  Distorted cells, a deformed state,
    per-node tractions; also on slabs and on a lattice that lies rotated over the box (cut along x)."""
    lo, hi = (0.0,) * dim, tuple(0.1 * r for r in reps)
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = 0.04 * 0.1 * np.random.default_rng(11).standard_normal((nverts, dim))
    roles = [O.FACE_CLAMPED] + [O.FACE_INTERFACE] * 5
    res = {}
    for correct in (0, 1):
        P = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, correct_face_F=correct), perturb)
        G = M.Context(dim=dim, degree=p, reps=reps, lo=lo, hi=hi, face_role=roles, perturb=perturb, slabs=slabs, cut_axis=cut_axis)
        G.set_tuning("correct_face_F", correct)
        _randomise_state(P, G, seed=5, amp_u=0.05)
        t = 1e3 * np.random.default_rng(3).standard_normal((len(P.interface_nodes), dim))
        P.set_interface_traction(t)
        G.set_interface_traction(t)
        P.update_acceleration()
        G.update_acceleration()
        P.assemble()
        G.assemble()
        res[correct] = (G.get(M.V_RHS), P.vec(O.V_RHS).copy())
        assert _relmax(res[correct][0], res[correct][1]) < TOL_ASM
        G.close()
    assert _relmax(res[1][1], res[0][1]) > 1e-4  # the two pull-backs really differ on a deformed mesh


@pytest.mark.parametrize("dim,p,reps", [(3, 2, (3, 3, 3)), (3, 1, (4, 4, 4)), (2, 3, (6, 3))])
def test_spmv_matches_reference_matrix(dim, p, reps):
    P, G = _pair(dim, p, reps, perturb_amp=0.05, seed=5)
    _randomise_state(P, G, seed=6)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    G.assemble()
    x = np.random.default_rng(4321).standard_normal(P.n)
    y_ref = P.csr() @ x
    y = G.spmv(x)
    assert _relmax(y, y_ref) < 1e-13
    # symmetry through the kernel: x.Ky == y.Kx
    z = np.random.default_rng(99).standard_normal(P.n)
    assert abs(z @ G.spmv(x) - x @ G.spmv(z)) / abs(z @ y) < 1e-12


def test_element_kernel_variants_agree():
    """the 3D Q2 element kernels -- 0: the sum-factorised default (assemble_q2sf: 45 coefficient fields, branch-free scatter),
    3: the same kernel as of round 4 (81 fields), 4-8: the A/B combinations of round 5 (pipelined contraction, prologue
    priority, parts of the default alone), 9: the node-pair kernel every other element uses, 1 / 2: its quadrature chunk
    sizes -- assemble the same tangent and residual; so does the residual-only pass of each family.  Constrained faces
    and a perturbed mesh: the masking instantiation of the scatter runs beside the plain one."""
    reps = (3, 3, 2)
    nverts = int(np.prod([r + 1 for r in reps]))
    perturb = 0.02 * np.random.default_rng(3).standard_normal((nverts, 3))
    G = M.Context(dim=3, degree=2, reps=reps, hi=(0.6, 0.6, 0.4), perturb=perturb, body_force=(0.0, -9.81, 0.0))
    rng = np.random.default_rng(4)
    G.set(M.V_U, 2e-3 * rng.standard_normal(G.n) * ~G.constrained)
    G.set(M.V_V_OLD, 0.1 * rng.standard_normal(G.n))
    G.set_interface_traction(1e3 * rng.standard_normal((len(G.interface()[0]), 3)))
    G.update_acceleration()
    x = rng.standard_normal(G.n)
    ref = None
    # (3-8 and the two-kernel forms below are A/B instantiations of the experiments build: make EXPERIMENTS=1 + MI_LIB)
    exp = G.get_tuning("experiments") == 1
    for v in (0, 3, 4, 5, 6, 7, 8, 9, 1, 2) if exp else (0, 9, 1, 2):
        G.set_tuning("asm_variant", v)
        rn = G.assemble()
        y, r = G.spmv(x), G.get(M.V_RHS)
        if v in (0, 9):  # the residual-only pass of the family: bit-identical with its full kernel
            assert G.assemble_residual() == rn and np.array_equal(G.get(M.V_RHS), r)
        if ref is None:
            ref = (y, r, rn)
            continue
        assert np.abs(y - ref[0]).max() / np.abs(ref[0]).max() < 1e-13, v
        assert np.abs(r - ref[1]).max() / np.abs(ref[1]).max() < 1e-13, v
        assert abs(rn - ref[2]) / ref[2] < 1e-13
    # round 6, the default where point records exist: the tangent in two kernels -- the point pass (the residual kernel,
    # which also writes F, J^(-2/3), 1/J of every point) and the tangent FROM those records (every wave of the workgroup
    # recomputes the material response, a quarter of the fields each); against the fused kernel ("asm_split" 0)
    G.set_tuning("asm_variant", 0)
    G.set_tuning("element_tangents", 2)
    K = {}
    if not exp:
        with pytest.raises(M.MiError):
            G.set_tuning("asm_split", 1)
        with pytest.raises(M.MiError):
            G.set_tuning("asm_variant", 5)
        return
    for split in (0, 1, 2):
        G.set_tuning("asm_split", split)
        rn = G.assemble()
        K[split] = (G.csr().data.copy(), G.get(M.V_RHS), rn, G.spmv(x))
        G.set_tuning("spmv_variant", 4)  # the matrix-free product from the records either kernel left
        assert np.abs(G.spmv(x) - K[split][3]).max() / np.abs(K[split][3]).max() < 1e-13
        G.set_tuning("spmv_variant", 3)
    for split in (1, 2):
        assert np.abs(K[split][0] - K[0][0]).max() / np.abs(K[0][0]).max() < 1e-13
        assert np.array_equal(K[split][1], K[0][1]) and K[split][2] == K[0][2]  # the residual: the same instruction stream
        assert np.abs(K[split][3] - ref[0]).max() / np.abs(ref[0]).max() < 1e-13


def _diag_blocks_of(K, dim):
    """[n_nodes, dim, dim] diagonal blocks of a scipy CSR matrix"""
    nn = K.shape[0] // dim
    D, base = np.zeros((nn, dim, dim)), np.arange(nn) * dim
    for i in range(dim):
        for j in range(dim):
            D[:, i, j] = np.asarray(K[base + i, base + j]).ravel()
    return D


@pytest.mark.parametrize("perturb_amp,reps,slabs", [(0.0, (4, 3, 5), 1), (0.05, (4, 3, 5), 1), (0.05, (3, 3, 7), 2), (0.0, (2, 1, 1), 1)])
def test_matrix_free_fine_level_against_the_oracle(perturb_amp, reps, slabs):
    """round 6, tuning "fine_level" 1: nothing of the fine tangent is assembled.  What the level keeps of it -- the residual,
    the operator (every product through mf_spmv from the point records, constrained rows / columns and their diagonal
    rule included) and the nodes' diagonal blocks formed from the records (mf_diag; |K_e(i,i)| on the diagonal of a
    constrained dof, its row and column dropped: deal.II's distribute_local_to_global as the scatter applies it
    [REF nonlinear_elasticity.cc:760-774, 1011-1023]) -- against the oracle's assembled system; boxes and distorted cells,
    clamped and z-clamped faces, one slab and two; the converged Newton update of a Jacobi-PCG on that operator too."""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    P, G = _pair(3, 2, reps, perturb_amp=perturb_amp, seed=21, roles=roles, body_force=(0.0, -9.81, 2.0), slabs=slabs)
    G.set_tuning("fine_level", 1)
    with pytest.raises(M.MiError):  # no tangent yet: a product has nothing to multiply with
        G.spmv(np.ones(G.n))
    rng = np.random.default_rng(23)
    for rnd in range(2):
        _randomise_state(P, G, seed=22 + rnd)
        P.update_acceleration()
        P.assemble()
        G.update_acceleration()
        rn = G.assemble()
        K = P.csr()
        assert abs(rn - P.residual_norm()) / P.residual_norm() < 1e-12
        assert _relmax(G.get(M.V_RHS), P.vec(O.V_RHS)) < TOL_ASM
        D_o, D_g = _diag_blocks_of(K, 3), G.diagonal_blocks()
        assert _relmax(D_g, D_o) < TOL_ASM
        cons = P.constrained.reshape(-1, 3)
        for c in range(3):  # constrained dofs: unit row / column up to the kept diagonal, exactly
            assert np.all(D_g[cons[:, c], c, (c + 1) % 3] == 0) and np.all(D_g[cons[:, c], (c + 2) % 3, c] == 0)
        x = rng.standard_normal(G.n)
        y = G.spmv(x)
        assert _relmax(y, K @ x) < 1e-13
        assert np.array_equal(G.spmv(x), y)  # fixed summation order
        assert G.assemble_residual() == rn and _relmax(G.get(M.V_RHS), P.vec(O.V_RHS)) < TOL_ASM
        assert np.array_equal(G.spmv(x), y)  # the residual-only pass left the tangent's records alone
    with pytest.raises(M.MiError):
        G.csr()
    G.set_tuning("precond", 0)
    rc, its, res = G.cg_solve(1e-12, 4 * G.n)
    rc_o, its_o, _ = P.solve_linear(O.SOLVER_CG_JACOBI, tol_lin=1e-12, max_it_mult=4.0)
    assert rc == 0 and rc_o == 0
    assert _relmax(G.get(M.V_NEWTON), P.vec(O.V_NEWTON)) < TOL_SOL
    # ... and back: the assembled level returns with the next assembly
    G.set_tuning("fine_level", 0)
    G.assemble()
    if slabs == 1:  # (matrix export: undecomposed meshes only)
        assert _relmax(G.csr().data, K.data) < TOL_ASM
    assert _relmax(G.spmv(x), K @ x) < 1e-13
    assert _relmax(G.diagonal_blocks(), D_o) < TOL_ASM


@pytest.mark.parametrize("perturb_amp,slabs", [(0.0, 1), (0.05, 1), (0.0, 3)])
def test_smoother_quadrature_3_operator_and_solves(perturb_amp, slabs):
    """round 6, tuning "smoother_quadrature" 3 (the library's default): the multigrid smoother's fine-level operator A' is the
    same tangent integrated with 3 x 3 x 3 Gauss points (mf_spmv27, two cells per wave, from records of its own), the CG's
    operator A keeps the assembly's 4 x 4 x 4 [REF nonlinear_elasticity.cc:74].  (1) On undeformed boxes the integrand is a
    polynomial the 3-point rule integrates exactly: A' x = A x to rounding -- the kernel, its records and its tables against the
    64-point product.  (2) On a deformed state (and on distorted cells) the two differ by the quadrature error of a smooth
    integrand only.  (3) A preconditioner-side choice: the multigrid-PCG converges to the same update with either rule, in
    the same number of iterations (+-1), on one slab and on three."""
    reps = (6, 5, 7)
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    P, G = _pair(3, 2, reps, perturb_amp=perturb_amp, seed=31, roles=roles, slabs=slabs)
    G.set_tuning("precond", 1)
    G.set_tuning("element_tangents", 2)
    assert G.get_tuning("smoother_quadrature") == 3
    rng = np.random.default_rng(32)
    x = rng.standard_normal(G.n)
    # (1) / (2): the operators
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.update_acceleration()
    G.assemble()
    assert G.get_tuning("smoother_quadrature_active") == 3
    y4 = G.spmv(x)
    G.set_tuning("spmv_as_smoother", 1)
    y3 = G.spmv(x)
    assert np.array_equal(G.spmv(x), y3)  # fixed summation order
    G.set_tuning("spmv_as_smoother", 0)
    if perturb_amp == 0.0:
        assert _relmax(y3, y4) < 1e-13
    else:
        assert 1e-9 < _relmax(y3, y4) < 2e-2
    # a smooth deformation: the quadrature error of a smooth integrand
    X = G.coords
    u = np.zeros((G.nnodes, 3))
    u[:, 1] = 0.02 * X[:, 0] ** 2
    u[:, 2] = 0.01 * np.sin(3 * X[:, 0]) * X[:, 1]
    u = u.reshape(-1) * ~G.constrained
    G.set(M.V_U, u)
    G.update_acceleration()
    G.assemble()
    G.set_tuning("spmv_as_smoother", 1)
    d = _relmax(G.spmv(x), G.spmv(x) * 0 + G.spmv(x))  # (same call twice: repeatable)
    assert d == 0.0
    y3 = G.spmv(x)
    G.set_tuning("spmv_as_smoother", 0)
    y4 = G.spmv(x)
    assert 1e-12 < _relmax(y3, y4) < 5e-3
    # the layout of the kernels' result slots (node-major | cell-major, the default under the 27-point rule | line-major) is a
    # permutation behind the gathers' index: the additions and their order are the same -- the same bits from either product
    for lay in (0, 2, 1):
        G.set_tuning("mf_slots_cell_major", lay)
        assert np.array_equal(G.spmv(x), y4)
        G.set_tuning("spmv_as_smoother", 1)
        assert np.array_equal(G.spmv(x), y3)
        G.set_tuning("spmv_as_smoother", 0)
    G.set_tuning("mf_slots_cell_major", -1)
    # (3): the solves
    sols = {}
    for q in (4, 3):
        G.set_tuning("smoother_quadrature", q)
        G.newton_begin_step()
        G.update_acceleration()
        G.assemble()
        assert G.get_tuning("smoother_quadrature_active") == q
        rc, its, res = G.cg_solve(1e-11, 4 * G.n)
        assert rc == 0 and 0 < its < 60
        sols[q] = (G.get(M.V_NEWTON), its)
    assert _relmax(sols[3][0], sols[4][0]) < 1e-8 and abs(sols[3][1] - sols[4][1]) <= 1


def test_smoother_quadrature_3_on_a_state_folded_between_the_assembly_points():
    """the 27-point rule looks at other points than the assembly: u_x = -1.05 x M(eta), M = 4 eta (1 - eta) in every cell's
    own eta along y, has det F = 1 - 1.05 M: 0.07 at the assembly's nearest Gauss points (eta = 0.33, 0.67), -0.05 at the
    3-point rule's eta = 0.5.  The assembly accepts the state (no det F <= 0 where the reference asserts,
    nonlinear_elasticity.cc:935); the smoother's records take the undeformed state at their folded points, so its operator
    stays finite (no cube root or reciprocal of a negative volume ratio enters the preconditioner) and the linear solve ends
    the way it does with the 64-point smoother (on THIS state the tangent itself is indefinite: no convergence either way)."""
    reps = (3, 3, 3)
    out = {}
    for q in (3, 4):
        G = M.Context(dim=3, degree=2, reps=reps)
        G.set_tuning("precond", 1)
        G.set_tuning("element_tangents", 2)
        G.set_tuning("smoother_quadrature", q)
        X = G.coords
        eta = (X[:, 1] * reps[1]) % 1.0
        u = np.zeros((G.nnodes, 3))
        u[:, 0] = -1.05 * X[:, 0] * 4.0 * eta * (1.0 - eta)
        G.set(M.V_U, u.reshape(-1) * ~G.constrained)
        G.set_interface_traction((0.0, -2e3, 0.0))
        G.newton_begin_step()
        G.update_acceleration()
        assert np.isfinite(G.assemble())
        assert G.get_tuning("smoother_quadrature_active") == q
        rng = np.random.default_rng(77)
        G.set_tuning("spmv_as_smoother", 1)
        for _ in range(4):
            x = rng.standard_normal(G.n) * ~G.constrained
            y = G.spmv(x)
            assert np.all(np.isfinite(y)) and x @ y > 0.0
        G.set_tuning("spmv_as_smoother", 0)
        rc, its, res = G.cg_solve(1e-10, 4 * G.n)
        assert np.isfinite(res) and np.all(np.isfinite(G.get(M.V_NEWTON)))
        out[q] = rc
        G.close()
    assert out[3] == out[4]


def test_fp32_smoother_records_follow_the_kernel_that_wrote_them():
    """only the sum-factorised kernel writes the fp32 point records; with the node-pair kernel ("asm_variant" 9)
    the opt-in fp32 smoother product must fall back to the fp64 records instead of multiplying with stale / uninitialised
    fp32 ones -- the preconditioned solve converges to the same update either way"""
    reps = (6, 6, 6)
    G = M.Context(dim=3, degree=2, reps=reps)
    G.set_tuning("precond", 1)
    G.set_tuning("element_tangents", 2)
    G.set_tuning("smoother_precision", 32)
    G.set_interface_traction((0.0, -2e3, 0.0))
    ref = None
    for v in (9, 0):
        G.set_tuning("asm_variant", v)
        G.newton_begin_step()
        G.update_acceleration()
        G.assemble()
        rc, its, res = G.cg_solve(1e-10, 2 * G.n)
        assert rc == 0 and 0 < its < 40, (v, its)
        du = G.get(M.V_NEWTON)
        assert np.all(np.isfinite(du))
        if ref is None:
            ref = (du, its)
        else:
            assert _relmax(du, ref[0]) < 1e-7 and abs(its - ref[1]) <= 2


@pytest.mark.parametrize("form", [1, 2])
def test_element_tangent_product_matches_the_assembled_matrix(form):
    """the multigrid smoother's operator on big undecomposed 3D Q2 meshes, in both unassembled forms: (1) the masked
    element tangents stored by the assembly (lower-triangle node-pair blocks) multiplied cell by cell, (2) the
    matrix-free product from the quadrature-point records of the assembly (sum factorisation) -- the same product as
    the assembled matrix, including constrained rows/columns and their diagonal rule; deterministic; follows every
    new tangent"""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    P, G = _pair(3, 2, (4, 3, 5), perturb_amp=0.05, seed=21, roles=roles)
    G.set_tuning("element_tangents", form)
    _randomise_state(P, G, seed=22)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    G.assemble()
    rng = np.random.default_rng(23)
    for _ in range(2):
        x = rng.standard_normal(G.n)
        y_ref = P.csr() @ x
        G.set_tuning("spmv_variant", 3)
        y_sell = G.spmv(x)
        G.set_tuning("spmv_variant", 4)
        y_ebe = G.spmv(x)
        assert _relmax(y_sell, y_ref) < 1e-13 and _relmax(y_ebe, y_ref) < 1e-13
        assert np.array_equal(G.spmv(x), y_ebe)  # fixed summation order
        _randomise_state(P, G, seed=24)  # a new tangent: the element blocks follow
        P.update_acceleration()
        P.assemble()
        G.update_acceleration()
        G.assemble()


def test_matrix_free_product_in_one_launch_is_bitwise_the_coloured_update():
    """mf_spmv in ONE launch (every cell stores into its own slots, mf_gather sums the slots of a node in processing
    order) performs the additions of the colour-by-colour update of y in the same order: identical bits"""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]
    for amp in (0.05, 0.0):  # distorted cells (Jacobian per point) and axis-parallel boxes (precomputed geometry)
        P, G = _pair(3, 2, (5, 4, 3), perturb_amp=amp, seed=31, roles=roles)
        G.set_tuning("element_tangents", 2)
        _randomise_state(P, G, seed=32)
        G.update_acceleration()
        G.assemble()
        x = np.random.default_rng(33).standard_normal(G.n)
        G.set_tuning("spmv_variant", 4)
        assert G.get_tuning("mf_single_launch") == 1
        y1 = G.spmv(x)
        G.set_tuning("mf_single_launch", 0)
        assert G.get_tuning("mf_single_launch") == 0
        y0 = G.spmv(x)
        assert np.array_equal(y0, y1)
        G.set_tuning("spmv_variant", 3)
        assert _relmax(y1, G.spmv(x)) < 1e-13


def test_exact_coarsest_level_solve_against_the_polynomial():
    """the dense inverse on the coarsest multigrid level (2^3 cells) against round 1's degree-12 polynomial on a one-cell
    level: a preconditioner at least as good (iterations) and the same converged solution"""
    res = {}
    for dense in (1, 0):
        G = M.Context(dim=3, degree=2, reps=(12, 10, 8))
        if not dense:  # (keys of the hierarchy's shape: set before "precond" 1 builds it, or it is rebuilt)
            G.set_tuning("mg_dense", 0)
            G.set_tuning("mg_coarsest", 1)
        G.set_tuning("precond", 1)
        G.set_interface_traction((0.0, -2e3, 0.0))
        G.newton_begin_step()
        G.update_acceleration()
        G.assemble()
        rc, its, r = G.cg_solve(rel_tol=1e-10)
        assert rc == 0
        res[dense] = (its, G.get(M.V_NEWTON))
        G.close()
    assert res[1][0] <= res[0][0] + 1 and 0 < res[1][0] < 40
    assert _relmax(res[1][1], res[0][1]) < 1e-8


def test_stalled_multigrid_solve_heals_itself():
    """a smoother interval that ends below lambda_max amplifies the top modes and the preconditioned CG crawls (what
    happened on the 120^3 mesh with a 15-iteration eigenvalue estimate).  mi_cg_solve gives a multigrid solve 300
    iterations, then estimates the eigenvalues of every level from scratch and continues: the solve ends converged
    instead of iterating on to dofs x multiplier"""
    G = M.Context(dim=3, degree=2, reps=(12, 12, 12))
    G.set_tuning("precond", 1)
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.newton_begin_step()
    G.update_acceleration()
    G.assemble()
    rc, its_ok, _ = G.cg_solve(rel_tol=1e-10)
    assert rc == 0 and its_ok < 40
    x_ok = G.get(M.V_NEWTON)
    G.set(M.V_NEWTON, np.zeros(G.n))
    G.set_tuning("mg_scale_lmax_percent", 45)  # spoil the estimates: Chebyshev on [.., 0.45 lambda_max]
    rc, its, _ = G.cg_solve(rel_tol=1e-10)
    assert rc == 0 and 300 < its < 300 + 3 * its_ok + 10, its
    assert _relmax(G.get(M.V_NEWTON), x_ok) < 1e-7
    G.set(M.V_NEWTON, np.zeros(G.n))
    rc, its, _ = G.cg_solve(rel_tol=1e-10)  # healed for good
    assert rc == 0 and abs(its - its_ok) <= 2


def test_coarse_operators_are_kept_over_time_steps_and_refreshed_on_demand():
    """"mg_refresh_every": the preconditioner's coarse operators (levels >= 1) are rebuilt at every k-th time step --
    same solutions at a tight tolerance whatever k -- and earlier when a solve needs a quarter more iterations than
    the first solve after the last rebuild (here: after the eigenvalue estimates were spoiled)"""
    def run(every, spoil_at=None):
        G = M.Context(dim=3, degree=2, reps=(16, 16, 16))  # 107,811 dofs: multigrid is the default
        assert G.get_tuning("precond") == 1 and G.get_tuning("mg_refresh_every") == 8
        G.set_tuning("mg_refresh_every", every)
        G.reset_timings()
        its, refreshes = [], []
        for k in range(4):
            if k == spoil_at:
                G.set_tuning("mg_scale_lmax_percent", 60)
            G.set_interface_traction((0.0, -2e3 * (k + 1), 0.0))
            rc, info = G.newmark_step(tol_lin=1e-10, max_it_mult=1.0)
            assert rc == 0 and info.converged == 1
            its.append([int(info.lin_its[i]) for i in range(info.newton_iterations)])
            refreshes.append(G.get_tuning("count_mg_refresh"))
        u = G.get(M.V_U)
        G.close()
        return its, refreshes, u

    its1, r1, u1 = run(1)
    assert r1 == [1, 2, 3, 4]
    its2, r2, u2 = run(2)
    assert r2 == [1, 1, 2, 2] and _relmax(u2, u1) < 1e-8
    itsn, rn, un = run(1000)
    assert rn == [1, 1, 1, 1] and _relmax(un, u1) < 1e-8
    assert max(max(s) for s in itsn) <= max(max(s) for s in its1) + 2  # (the lagged operators precondition as well)
    itss, rs, us = run(1000, spoil_at=2)
    assert rs[1] == 1 and rs[2] == 2, (rs, itss)  # the slow solve of step 3 asked for the rebuild, in that step
    assert max(itss[2]) > max(itsn[2]) + 2 and max(itss[3]) <= max(itsn[3]) + 2, (itss, itsn)
    assert _relmax(us, u1) < 1e-8


def test_smoother_operator_choice_only_changes_the_preconditioner():
    """multigrid-PCG with the smoother on the element tangents vs on the assembled matrix: the same operator up to
    rounding, so the same iteration counts and the same converged solution"""
    G = M.Context(dim=3, degree=2, reps=(24, 24, 24))  # 352,947 dofs, 117,649 nodes: above the 100k-node switch
    G.set_interface_traction((0.0, -2e3, 0.0))
    assert G.get_tuning("smoother_operator_active") == 2  # default: matrix-free from the quadrature-point records
    res = {}
    for op in (2, 1, 0):
        G.set_tuning("smoother_operator", op)
        G.set(M.V_NEWTON, np.zeros(G.n))
        G.newton_begin_step()
        G.update_acceleration()
        G.assemble()
        rc, its, r = G.cg_solve(rel_tol=1e-10)
        assert rc == 0
        res[op] = (its, G.get(M.V_NEWTON))
        assert G.get_tuning("smoother_operator_active") == op
    for op in (1, 2):
        assert abs(res[0][0] - res[op][0]) <= 1 and 0 < res[op][0] < 40
        assert _relmax(res[op][1], res[0][1]) < 1e-8


@pytest.mark.parametrize("form", [1, 2])
def test_cg_operator_choice_gives_the_same_solve(form):
    """A/B switch "cg_operator": the CG's own product on the unassembled form the smoother uses (element tangents or
    quadrature-point records; p.q by a separate reduction, no sliced-ELL copy) instead of the assembled matrix -- same
    iteration count (+-1), same solution"""
    G = M.Context(dim=3, degree=2, reps=(24, 24, 24))
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.set_tuning("smoother_operator", form)
    res = {}
    for op in (0, 1):
        G.set_tuning("cg_operator", op)
        G.set(M.V_NEWTON, np.zeros(G.n))
        G.newton_begin_step()
        G.update_acceleration()
        G.assemble()
        rc, its, r = G.cg_solve(rel_tol=1e-10)
        assert rc == 0
        res[op] = (its, G.get(M.V_NEWTON))
    assert abs(res[0][0] - res[1][0]) <= 1 and _relmax(res[1][1], res[0][1]) < 1e-8
    rc, info = G.newmark_step(tol_lin=1e-8)  # and through a whole step
    assert rc == 0 and info.converged == 1


def test_spmv_kernel_variants_agree():
    """sliced-ELL (production) and block-CSR (cross-check) kernels on the same matrix, several grids"""
    P, G = _pair(3, 2, (5, 4, 3), perturb_amp=0.05, seed=11)
    _randomise_state(P, G, seed=12)
    G.update_acceleration()
    G.assemble()
    x = np.random.default_rng(7).standard_normal(P.n)
    ys = {}
    for variant in (3, 1):
        for grid in (1, 7, 64):
            G.set_tuning("spmv_variant", variant)
            G.set_tuning("spmv_grid", grid)
            ys[(variant, grid)] = G.spmv(x)
    ref = ys[(3, 64)]
    for k, y in ys.items():
        assert _relmax(y, ref) < 1e-14, k
    P.update_acceleration()
    P.assemble()
    assert _relmax(ref, P.csr() @ x) < 1e-13


@pytest.mark.parametrize("dim,p,reps", [(3, 2, (3, 3, 3)), (2, 2, (18, 3))])
def test_cg_solution_and_iteration_count(dim, p, reps):
    """Jacobi-PCG on the device vs the oracle's Jacobi-PCG: same stopping rule, same iterate count (+-1 from
    summation order), converged solution equal to the direct solve"""
    P, G = _pair(dim, p, reps, perturb_amp=0.03, seed=7)
    _randomise_state(P, G, seed=8)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    G.assemble()
    # loose tolerance: iterate-level agreement (Jacobi on both sides)
    G.set_tuning("precond", 0)
    P.vec(O.V_NEWTON)[:] = 0
    rc_o, its_o, res_o = P.solve_linear(O.SOLVER_CG_JACOBI, tol_lin=1e-6, max_it_mult=1.0)
    rc_g, its_g, res_g = G.cg_solve(rel_tol=1e-6)
    assert rc_o == 0 and rc_g == 0
    assert abs(its_g - its_o) <= 1
    assert res_g <= 1e-6 * np.linalg.norm(P.vec(O.V_RHS))
    x_g = G.get(M.V_NEWTON)
    assert _relmax(x_g, P.vec(O.V_NEWTON)) < 1e-5
    assert np.all(x_g[P.constrained] == 0)
    # tight tolerance with warm start from the loose solution: solver-independent regime
    rc_g, its2, _ = G.cg_solve(rel_tol=1e-12)
    assert rc_g == 0 and its2 > 0
    P.vec(O.V_NEWTON)[:] = 0
    assert P.solve_linear(O.SOLVER_DIRECT)[0] == 0
    assert _relmax(G.get(M.V_NEWTON), P.vec(O.V_NEWTON)) < TOL_SOL
    # already converged: zero iterations, like SolverControl's initial check
    rc_g, its3, _ = G.cg_solve(rel_tol=1e-6)
    assert rc_g == 0 and its3 == 0


@pytest.mark.parametrize("dim,p,reps", [(3, 2, (6, 5, 4)), (3, 1, (9, 8, 7)), (2, 2, (18, 6)), (2, 4, (6, 4))])
def test_multigrid_preconditioner(dim, p, reps):
    """V-cycle preconditioned CG: same solution as the direct solve, far fewer iterations than Jacobi, repeatable"""
    P, G = _pair(dim, p, reps, perturb_amp=0.03, seed=21)
    _randomise_state(P, G, seed=22)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    G.assemble()
    G.set_tuning("precond", 0)
    rc, its_jacobi, _ = G.cg_solve(rel_tol=1e-10)
    assert rc == 0
    x_jacobi = G.get(M.V_NEWTON)
    G.set(M.V_NEWTON, np.zeros(P.n))
    G.set_tuning("precond", 1)
    rc, its_mg, res = G.cg_solve(rel_tol=1e-10)
    assert rc == 0 and res <= 1e-10 * np.linalg.norm(P.vec(O.V_RHS))
    assert its_mg * (3 if its_jacobi > 100 else 2) < its_jacobi, (its_mg, its_jacobi)
    x_mg = G.get(M.V_NEWTON)
    assert _relmax(x_mg, x_jacobi) < 1e-7
    P.vec(O.V_NEWTON)[:] = 0
    assert P.solve_linear(O.SOLVER_DIRECT if P.n < 4000 else O.SOLVER_CG_SSOR, tol_lin=1e-13, max_it_mult=2.0)[0] == 0
    assert _relmax(x_mg, P.vec(O.V_NEWTON)) < 1e-7
    G.set(M.V_NEWTON, np.zeros(P.n))
    rc, its2, _ = G.cg_solve(rel_tol=1e-10)
    assert its2 == its_mg and np.array_equal(G.get(M.V_NEWTON), x_mg)  # deterministic


def test_cg_reports_non_convergence():
    P, G = _pair(3, 1, (3, 3, 3))
    _randomise_state(P, G, seed=9)
    G.update_acceleration()
    G.assemble()
    G.set_tuning("precond", 0)
    rc, its, res = G.cg_solve(rel_tol=1e-14, max_it=3)
    assert rc == M.MI_ENOCONV_LIN and its == 3 and res > 0


@pytest.mark.parametrize("path", ["small", "jacobi", "multigrid"])
def test_cg_breakdown_is_reported_at_once(path):
    """a NaN right-hand side (or an indefinite operator) must end the solve immediately, as deal.II's SolverControl
    does, instead of running to n_dofs * multiplier iterations"""
    P, G = _pair(3, 1, (6, 6, 6) if path == "multigrid" else (3, 3, 3))
    _randomise_state(P, G, seed=9)
    G.set_tuning("precond", 1 if path == "multigrid" else 0)
    G.set_tuning("small_cg", 1 if path == "small" else 0)
    G.update_acceleration()
    G.assemble()
    rhs = G.get(M.V_RHS)
    rhs[7] = np.nan
    G.set(M.V_RHS, rhs)
    rc, its, res = G.cg_solve(rel_tol=1e-10)
    assert rc == M.MI_ENOCONV_LIN and its <= 1
    assert b"broke down" in M.lib().mi_last_error(G.h)
    # the context stays usable: a clean assembly + solve converges afterwards
    G.set(M.V_NEWTON, np.zeros(G.n))
    G.assemble()
    rc, its, res = G.cg_solve(rel_tol=1e-10)
    assert rc == 0 and its > 1


@pytest.mark.parametrize("scenario,dim,p", [("FSI3", 2, 1), ("FSI3", 2, 3), ("PF", 2, 2), ("FSI3", 3, 1), ("PF", 3, 2),
                                            ("FSI3", 3, 3), ("PF", 3, 4)])
def test_newmark_steps_interface_displacement(scenario, dim, p):
    """the reference's own geometries: 4 Newmark steps under a ramped traction; interface displacements matched
    by vertex coordinate against the oracle run with the reference solver configuration (CG+SSOR)"""
    d = O.scenario_desc(scenario, dim, degree=p)
    P = O.Problem(d)
    G = M.Context(dim=dim, degree=p, reps=tuple(d.reps)[:dim], lo=tuple(d.lo)[:dim], hi=tuple(d.hi)[:dim],
                  face_role=list(d.face_role))
    ids, xyz = G.interface()
    assert np.array_equal(ids, P.interface_nodes)
    tvec = np.array([0.0, -40.0, 0.0])[:dim] if scenario == "FSI3" else np.array([30.0, 0.0, 0.0])[:dim]
    for step in range(1, 5):
        t = tvec * step / 4.0
        P.set_interface_traction(t)
        G.set_interface_traction(t)
        rc_o, info_o = P.newmark_step(O.SOLVER_CG_SSOR, tol_lin=1e-12, max_it_mult=2.0)
        rc_g, info_g = G.newmark_step(tol_lin=1e-12, max_it_mult=2.0)
        assert rc_o == 0 and rc_g == 0 and info_g.converged == 1
        assert info_g.newton_iterations == info_o.newton_iterations
        assert info_g.assemblies == info_g.newton_iterations + 1
        u_o = P.vec(O.V_U).reshape(-1, dim)[ids]
        u_g = G.get_interface_displacement()
        assert np.abs(u_g - u_o).max() / np.abs(u_o).max() < TOL_SOL
    for k in (M.V_U, M.V_V, M.V_A, M.V_U_OLD, M.V_V_OLD, M.V_A_OLD):
        assert _relmax(G.get(k), P.vec(k)) < 1e-6  # v, a amplify displacement differences by 1/dt, 1/dt^2


def test_newton_table_values_follow_reference_logic():
    P, G = _pair(3, 2, (3, 2, 2))
    t = (0.0, -2e3, 0.0)
    P.set_interface_traction(t)
    G.set_interface_traction(t)
    G.set_tuning("precond", 0)
    G.set_tuning("cg_warm_start", 1)  # the reference's start vector for the later solves of a step (:419, :472-473)
    rc_o, io = P.newmark_step(O.SOLVER_CG_JACOBI, tol_lin=1e-10)
    rc_g, ig = G.newmark_step(tol_lin=1e-10)
    assert rc_o == 0 and rc_g == 0
    assert (ig.newton_iterations, ig.assemblies, ig.converged) == (io.newton_iterations, io.assemblies, 1)
    assert abs(ig.res_abs - io.res_abs) <= 1e-6 * max(io.res_abs, 1e-9) + 1e-9
    assert abs(ig.lin_its_total - io.lin_its_total) <= ig.newton_iterations
    # the default start vector (zero for every solve): same Newton table, same state, fewer linear iterations
    P2, G2 = _pair(3, 2, (3, 2, 2))
    G2.set_interface_traction(t)
    G2.set_tuning("precond", 0)
    rc_z, iz = G2.newmark_step(tol_lin=1e-10)
    assert rc_z == 0 and (iz.newton_iterations, iz.assemblies, iz.converged) == (ig.newton_iterations, ig.assemblies, 1)
    assert iz.lin_its_total < ig.lin_its_total
    assert _relmax(G2.get(M.V_U), G.get(M.V_U)) < 1e-8
    # too few Newton iterations -> the reference's "No convergence in nonlinear solver!"
    rc, _ = G.newmark_step(max_it_nr=1, check=False)
    assert rc == M.MI_ENOCONV_NR


def test_start_vector_from_the_previous_time_step():
    """"cg_warm_start" 2 (what the executable and bench.py set): the j-th linear solve of a step starts from the solution
    of the j-th solve of the previous step -- same stopping rule, same Newton table, same states as the zero start to
    the solver tolerance, fewer iterations under a smoothly changing load; 3: extrapolated over two steps.  The history
    is part of mi_state_save / mi_state_restore: a restored run repeats the saved one bit by bit"""
    runs = {}
    for mode in (0, 2, 3):
        _, G = _pair(3, 2, (4, 3, 2))
        G.set_tuning("precond", 0)
        G.set_tuning("cg_warm_start", mode)
        its, table = [], []
        for k in range(6):
            G.set_interface_traction((0.0, -300.0 * (k + 1), 0.0))
            rc, info = G.newmark_step(tol_lin=1e-10)
            assert rc == 0 and info.converged == 1
            its.append(info.lin_its_total)
            table.append((info.newton_iterations, info.assemblies))
        runs[mode] = (its, table, G.get(M.V_U), G.get(M.V_A), G)
    for mode in (2, 3):
        assert runs[mode][1] == runs[0][1]
        assert runs[mode][0][0] == runs[0][0][0]  # no history in the first step
        assert sum(runs[mode][0][1:]) < sum(runs[0][0][1:]), (runs[mode][0], runs[0][0])
        assert _relmax(runs[mode][2], runs[0][2]) < 1e-8 and _relmax(runs[mode][3], runs[0][3]) < 1e-6
    G = runs[2][4]
    G.state_save()
    G.set_interface_traction((0.0, -2500.0, 0.0))
    G.newmark_step(tol_lin=1e-10)
    first = G.get(M.V_U)
    G.state_restore()
    G.newmark_step(tol_lin=1e-10)
    assert np.array_equal(G.get(M.V_U), first)


def test_state_checkpoint_roundtrip():
    """implicit-coupling checkpoint (adapter.h:447-489): save, advance, restore, advance again -> same result"""
    _, G = _pair(2, 2, (6, 2))
    G.set_tuning("precond", 0)  # Jacobi-PCG is bitwise repeatable; the multigrid keeps a running eigenvector estimate
    G.set_interface_traction((0.0, -30.0))
    G.newmark_step(tol_lin=1e-12)
    G.state_save()
    saved = [G.get(k) for k in range(6)]
    G.set_interface_traction((0.0, -60.0))
    G.newmark_step(tol_lin=1e-12)
    after = G.get(M.V_U)
    assert np.abs(after - saved[0]).max() > 0
    G.state_restore()
    for k in range(6):
        assert np.array_equal(G.get(k), saved[k])
    G.newmark_step(tol_lin=1e-12)
    assert np.array_equal(G.get(M.V_U), after)  # deterministic kernels => bitwise repeatable
    G.set_tuning("precond", 1)
    G.state_restore()
    G.newmark_step(tol_lin=1e-12)
    assert np.abs(G.get(M.V_U) - after).max() / np.abs(after).max() < 1e-9


def test_rejects_bad_arguments():
    with pytest.raises(M.MiError) as e:
        M.Context(dim=3, degree=5, reps=(2, 2, 2))
    assert e.value.code == M.MI_EINVAL
    with pytest.raises(M.MiError):
        M.Context(dim=3, degree=1, reps=(2, 2, 2), nu=0.5)
    _, G = _pair(2, 1, (2, 2))
    buf = np.zeros(2)
    assert M.lib().mi_set_interface_traction(G.h, 1, M._dp(buf)) == M.MI_EINVAL
    assert b"interface nodes" in M.lib().mi_last_error(G.h)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,p,reps", [(3, 2, (3, 3, 3)), (2, 3, (4, 3)), (3, 1, (3, 2, 2))])
def test_inverted_element_is_reported(dim, p, reps):
    """the reference asserts det F > 0 at every quadrature point (nonlinear_elasticity.cc:935, debug builds); the device
    reports the same condition by name instead of iterating on a folded mesh"""
    G = M.Context(dim=dim, degree=p, reps=reps)
    G.set_interface_traction((0.0,) * dim)
    G.update_acceleration()
    assert np.isfinite(G.assemble())  # a regular state assembles
    x = G.coords
    u = np.zeros((G.nnodes, dim))
    u[:, 0] = -2.5 * x[:, 0]  # F_xx = 1 - 2.5 < 0 everywhere
    G.set(M.V_U, (u * (~G.constrained.reshape(-1, dim))).ravel())
    with pytest.raises(M.MiError, match="inverted element"):
        G.assemble()
    G.set(M.V_U, np.zeros(G.n))  # the context survives
    assert np.isfinite(G.assemble())


# ---------------------------------------------------------------------------------------------------------------
# "Solver type = Direct": banded Cholesky on the device (mi_direct_solve) against the oracle's direct solve
DIRECT_CASES = [("FSI3", 2, 1), ("FSI3", 2, 2), ("FSI3", 2, 3), ("PF", 2, 2), ("PF", 3, 2), ("FSI3", 3, 1)]


def _scenario_pair(scenario, dim, p):
    d = O.scenario_desc(scenario, dim, degree=p)
    P = O.Problem(d)
    G = M.Context(dim=dim, degree=p, reps=tuple(d.reps)[:dim], lo=tuple(d.lo)[:dim], hi=tuple(d.hi)[:dim],
                  face_role=list(d.face_role))
    return P, G


@pytest.mark.parametrize("dim,p,reps", [(2, 3, (6, 5)), (2, 4, (5, 3)), (2, 2, (40, 9)), (2, 1, (60, 25)), (2, 3, (3, 1)), (2, 1, (2, 2)),
                                        (3, 1, (9, 3, 2)), (3, 1, (5, 4, 4)), (2, 3, (4, 7)), (2, 1, (100, 54)), (2, 1, (30, 55)),
                                        (2, 1, (61, 70)), (2, 1, (62, 64)), (2, 1, (69, 75)), (2, 2, (18, 30)), (3, 1, (2, 15, 20)),
                                        (2, 1, (77, 60)), (2, 1, (79, 40)), (3, 2, (2, 2, 9))])
def test_direct_solver_over_band_widths(dim, p, reps):
    """round 4: factorisations of bands up to 112 dofs wide run with the active window of the matrix in LDS (band_cholesky_lds: a
    circular window of 128 rows, look-ahead, streamed rows), wider ones on the general kernel.  Meshes on both sides of the
    limit, windows that wrap many times (11 k dofs), last blocks and panels shorter than a block column, systems smaller than
    the window; round 5: bands of 113-160 dofs (the reference's 3D plate at degree 2: 152) keep the window and take the rows
    beyond it through memory -- half bandwidths 113, 127, 129, 143, 153, 158, 159 (1 to 47 rows beyond the window, on both
    sides of a whole block column of them), 163 and 182 on the general kernel: the direct solution against scipy's sparse LU of the exported tangent (an independent solver standing in for
    the reference's UMFPACK) [REF nonlinear_elasticity.cc:1192-1200]; constrained dofs come back as exact zeros (:1208)."""
    import ctypes as C
    import scipy.sparse.linalg as spla
    L = M.lib()
    L.mi_direct_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    G = M.Context(dim=dim, degree=p, reps=reps, hi=tuple(0.1 * r for r in reps))
    G.set_tuning("precond", 0)
    rng = np.random.default_rng(3)
    G.set(M.V_U, 0.01 * 0.1 / p * rng.standard_normal(G.n) * ~G.constrained)
    G.set_interface_traction(tuple([0.0, -2e3, 0.0][:dim]))
    G.update_acceleration()
    G.assemble()
    res = C.c_double(0)
    assert L.mi_direct_solve(G.h, C.byref(res)) == 0, L.mi_last_error(G.h)
    xd = G.get(M.V_NEWTON)
    K, b = G.csr(), G.get(M.V_RHS)
    xs = spla.spsolve(K.tocsc(), b)
    xs[G.constrained] = 0.0  # constraints.distribute: the exported rhs is zero there up to the constrained rows' own diagonal
    assert _relmax(xd, xs) < 1e-9
    assert np.all(xd[G.constrained] == 0.0)
    G.close()


@pytest.mark.parametrize("dim,p,reps", [(2, 3, (6, 5)), (2, 1, (100, 54)), (2, 1, (62, 64)), (2, 2, (18, 30)), (3, 1, (2, 15, 20)),
                                        (2, 1, (79, 40)), (2, 1, (300, 30)), (3, 2, (3, 2, 6)), (2, 1, (2, 2))])
def test_linear_model_substitutions_over_band_widths(dim, p, reps):
    """round 5: the linear model factorises its constant matrix in the first step and only substitutes afterwards
    (linear_elasticity.cc:479-497 solves with the matrix assembled once, :236); the substitutions run with x in LDS, the
    panel updates on the matrix cores (band_solve_lds) for systems up to 18,432 dofs and half bandwidths up to 240.  Meshes
    on both sides of both limits and of the window kernel's (half bandwidths 67, 113, 129, 153, 158, 163, 65 with 18,662 dofs,
    248, and a system smaller than a block column): three steps -- factorisation + substitution, then substitutions alone,
    with both right-hand-side paths -- against the oracle's banded LU."""
    import ctypes as C
    hi = tuple(0.1 * r for r in reps)
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_ZCLAMP]
    P = O.LinearProblem(O.make_desc(dim=dim, degree=p, reps=reps, hi=hi, face_role=roles, theta=0.6))
    G = M.Context(dim=dim, degree=p, reps=reps, hi=hi, face_role=roles)
    G.set_tuning("solver_type", 1)
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    assert L.mi_linear_setup(G.h, 0.6) == 0, L.mi_last_error(G.h)
    rng = np.random.default_rng(11)
    ids = P.interface_nodes
    for step in range(3):
        consistent = step != 1
        t = 50.0 * rng.standard_normal((len(ids), dim))
        P.vec(O.L_STRESS)[:] = 0
        for c in range(dim):
            P.vec(O.L_STRESS)[ids * dim + c] = t[:, c]
        G.set_interface_traction(t)
        assert P.step(O.SOLVER_DIRECT, consistent)[0] == 0
        its, res = C.c_int(0), C.c_double(0)
        assert L.mi_linear_step(G.h, int(consistent), 1e-12, G.n * 10, C.byref(its), C.byref(res)) == 0, L.mi_last_error(G.h)
        assert its.value == 1  # one "iteration" per direct solve
        for vo, vg in ((O.L_D, 0), (O.L_V, 2)):
            assert _relmax(G.get(vg), P.vec(vo)) < 1e-9, (step, vo)
    G.close()


@pytest.mark.parametrize("scenario,dim,p", DIRECT_CASES)
def test_direct_solver_matches_the_oracle_direct_solve(scenario, dim, p):
    """the reference's shipped default (parameters.prm:43, nonlinear_elasticity.cc:1192-1200) on its own geometries: one
    factorisation + substitution on the device = the oracle's banded LU = scipy's sparse LU of the exported matrix"""
    import scipy.sparse.linalg as spla
    P, G = _scenario_pair(scenario, dim, p)
    _randomise_state(P, G, seed=5 + p, amp_u=0.002)
    P.update_acceleration()
    P.assemble()
    G.update_acceleration()
    G.assemble()
    P.vec(O.V_NEWTON)[:] = 0.0
    rc_o, _, _ = P.solve_linear(O.SOLVER_DIRECT)
    assert rc_o == 0
    assert G.direct_solve() == 0
    x = G.get(M.V_NEWTON)
    assert _relmax(x, P.vec(O.V_NEWTON)) < 1e-10
    K, b = G.csr(), G.get(M.V_RHS)
    assert _relmax(x, spla.spsolve(K.tocsc(), b)) < 1e-10
    assert np.all(x[G.constrained] == 0.0)  # constraints.distribute (:1208)
    x2 = None
    for _ in range(2):  # bit-stable: the same factorisation twice
        assert G.direct_solve() == 0
        x2 = G.get(M.V_NEWTON) if x2 is None else x2
        assert np.array_equal(G.get(M.V_NEWTON), x2)


@pytest.mark.parametrize("scenario,dim,p", [("FSI3", 2, 3), ("PF", 3, 2)])
def test_newmark_steps_with_the_direct_solver(scenario, dim, p):
    """whole steps with solver_type = 1: Newton table and interface displacement as the oracle's direct steps"""
    P, G = _scenario_pair(scenario, dim, p)
    G.set_tuning("solver_type", 1)
    ids = P.interface_nodes
    for step in range(3):
        t = np.zeros(dim)
        t[1] = -50.0 * (step + 1)
        P.set_interface_traction(t)
        G.set_interface_traction(t)
        rc_o, io = P.newmark_step(O.SOLVER_DIRECT)
        rc, ig = G.newmark_step()
        assert rc == 0 and rc_o == 0 and ig.converged == 1
        assert (ig.newton_iterations, ig.assemblies) == (io.newton_iterations, io.assemblies)
        assert ig.lin_its_total == ig.newton_iterations  # one "iteration" per direct solve (:1198)
        u_o = P.vec(O.V_U).reshape(-1, dim)[ids]
        assert _relmax(G.get_interface_displacement(), u_o) < 1e-10


def test_direct_solver_refuses_what_it_cannot_factorise():
    """beyond ~3e8 flops of factorisation the entry point says so and the step falls back to the PCG at 1e-12"""
    P, G = _pair(3, 2, (12, 12, 12))
    G.set_interface_traction((0.0, -2e3, 0.0))
    G.update_acceleration()
    G.assemble()
    assert G.direct_solve() == M.MI_EINVAL and b"too large" in M.lib().mi_last_error(G.h)
    G.set_tuning("solver_type", 1)
    P.set_interface_traction((0.0, -2e3, 0.0))
    rc, info = G.newmark_step()
    rc_o, io = P.newmark_step(O.SOLVER_CG_JACOBI, tol_lin=1e-12, max_it_mult=2.0)
    assert rc == 0 and rc_o == 0 and info.converged == 1 and info.lin_its_total > info.newton_iterations
    assert _relmax(G.get(M.V_U), P.vec(O.V_U)) < TOL_SOL
