"""ctypes binding of oracle/liboracle.so (CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

FACE_CLAMPED, FACE_INTERFACE, FACE_ZCLAMP = 1, 7, 8

(V_U, V_U_OLD, V_V, V_V_OLD, V_A, V_A_OLD, V_STRESS, V_DELTA, V_NEWTON, V_RHS) = range(10)
(L_D, L_D_OLD, L_V, L_V_OLD, L_STRESS, L_STRESS_OLD, L_RHS) = range(7)
SOLVER_CG_SSOR, SOLVER_CG_JACOBI, SOLVER_DIRECT = 0, 1, 2


class Desc(C.Structure):
    _fields_ = [
        ("dim", C.c_int),
        ("degree", C.c_int),
        ("reps", C.c_int * 3),
        ("lo", C.c_double * 3),
        ("hi", C.c_double * 3),
        ("face_role", C.c_int * 6),
        ("mu", C.c_double),
        ("nu", C.c_double),
        ("rho", C.c_double),
        ("body_force", C.c_double * 3),
        ("beta", C.c_double),
        ("gamma", C.c_double),
        ("delta_t", C.c_double),
        ("theta", C.c_double),
        ("correct_face_F", C.c_int),
    ]


class StepInfo(C.Structure):
    _fields_ = [
        ("newton_iterations", C.c_int),
        ("assemblies", C.c_int),
        ("lin_its_total", C.c_int),
        ("converged", C.c_int),
        ("res_norm", C.c_double),
        ("res_abs", C.c_double),
        ("upd_norm", C.c_double),
        ("upd_abs", C.c_double),
        ("t_assemble", C.c_double),
        ("t_solve", C.c_double),
    ]


class Time(C.Structure):
    _fields_ = [
        ("timestep", C.c_uint),
        ("time_current", C.c_double),
        ("time_end", C.c_double),
        ("delta_t", C.c_double),
    ]


def build():
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("elasticity_oracle.cpp", "elasticity_oracle.h")
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_gauss_01.argtypes = [C.c_int, dp, dp]
        L.orc_feq_support_1d.argtypes = [C.c_int, dp]
        L.orc_lagrange_1d.argtypes = [C.c_int, C.c_double, dp, dp]
        L.orc_material.restype = C.c_double
        L.orc_material.argtypes = [C.c_int, C.c_double, C.c_double, dp, dp, dp]
        L.orc_cell_tangent_residual.argtypes = [C.POINTER(Desc), dp, dp, dp, dp, dp]
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Desc), dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_n_dofs", "orc_n_nodes", "orc_n_cells", "orc_n_interface_nodes"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_nnz.restype = C.c_long
        L.orc_nnz.argtypes = [C.c_void_p]
        L.orc_node_coords.restype = dp
        L.orc_node_coords.argtypes = [C.c_void_p]
        L.orc_csr_rowptr.restype = ip
        L.orc_csr_rowptr.argtypes = [C.c_void_p]
        L.orc_csr_col.restype = ip
        L.orc_csr_col.argtypes = [C.c_void_p]
        L.orc_csr_val.restype = dp
        L.orc_csr_val.argtypes = [C.c_void_p]
        L.orc_constrained.restype = C.POINTER(C.c_ubyte)
        L.orc_constrained.argtypes = [C.c_void_p]
        L.orc_interface_nodes.restype = ip
        L.orc_interface_nodes.argtypes = [C.c_void_p]
        L.orc_vec.restype = dp
        L.orc_vec.argtypes = [C.c_void_p, C.c_int]
        L.orc_update_acceleration.argtypes = [C.c_void_p]
        L.orc_assemble.argtypes = [C.c_void_p]
        L.orc_residual_norm.restype = C.c_double
        L.orc_residual_norm.argtypes = [C.c_void_p]
        L.orc_solve_linear.restype = C.c_int
        L.orc_solve_linear.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, ip, dp]
        L.orc_spmv.argtypes = [C.c_void_p, dp, dp]
        L.orc_newmark_step.restype = C.c_int
        L.orc_newmark_step.argtypes = [
            C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(StepInfo)]
        L.orc_linear_create.restype = C.c_void_p
        L.orc_linear_create.argtypes = [C.POINTER(Desc)]
        L.orc_linear_destroy.argtypes = [C.c_void_p]
        L.orc_linear_n_dofs.restype = C.c_int
        L.orc_linear_n_dofs.argtypes = [C.c_void_p]
        L.orc_linear_nnz.restype = C.c_long
        L.orc_linear_nnz.argtypes = [C.c_void_p]
        L.orc_linear_rowptr.restype = ip
        L.orc_linear_rowptr.argtypes = [C.c_void_p]
        L.orc_linear_col.restype = ip
        L.orc_linear_col.argtypes = [C.c_void_p]
        L.orc_linear_matrix.restype = dp
        L.orc_linear_matrix.argtypes = [C.c_void_p, C.c_int]
        L.orc_linear_node_coords.restype = dp
        L.orc_linear_node_coords.argtypes = [C.c_void_p]
        L.orc_linear_n_interface_nodes.restype = C.c_int
        L.orc_linear_n_interface_nodes.argtypes = [C.c_void_p]
        L.orc_linear_interface_nodes.restype = ip
        L.orc_linear_interface_nodes.argtypes = [C.c_void_p]
        L.orc_linear_constrained.restype = C.POINTER(C.c_ubyte)
        L.orc_linear_constrained.argtypes = [C.c_void_p]
        L.orc_linear_vec.restype = dp
        L.orc_linear_vec.argtypes = [C.c_void_p, C.c_int]
        L.orc_linear_step.restype = C.c_int
        L.orc_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, dp]
        L.orc_linear_step_tol.restype = C.c_int
        L.orc_linear_step_tol.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, ip, dp]
        L.orc_time_init.argtypes = [C.POINTER(Time), C.c_double, C.c_double]
        L.orc_time_increment.argtypes = [C.POINTER(Time)]
        L.orc_time_set_absolute.argtypes = [C.POINTER(Time), C.c_double]
        L.orc_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_desc(dim=3, degree=1, reps=(2, 2, 2), lo=(0, 0, 0), hi=(1, 1, 1), face_role=None, mu=0.5e6, nu=0.4,
              rho=1000.0, body_force=(0, 0, 0), beta=0.25, gamma=0.5, delta_t=0.005, theta=0.5, correct_face_F=0):
    d = Desc()
    d.dim, d.degree = dim, degree
    reps = tuple(reps) + (1,) * (3 - len(reps))
    lo = tuple(lo) + (0.0,) * (3 - len(lo))
    hi = tuple(hi) + (0.0,) * (3 - len(hi))
    if face_role is None:  # "Block": clamped x-, interface elsewhere
        face_role = [FACE_CLAMPED] + [FACE_INTERFACE] * 5
    for i in range(3):
        d.reps[i], d.lo[i], d.hi[i], d.body_force[i] = reps[i], lo[i], hi[i], body_force[i]
    for i in range(6):
        d.face_role[i] = face_role[i]
    d.mu, d.nu, d.rho = mu, nu, rho
    d.beta, d.gamma, d.delta_t, d.theta = beta, gamma, delta_t, theta
    d.correct_face_F = correct_face_F
    return d


# scenario -> (reps, lo, hi, face roles) for the reference's two geometries
# nonlinear_elasticity.cc:189-226, 265-278
def scenario_desc(scenario, dim, flap_location=0.0, **kw):
    if scenario == "FSI3":
        reps = (18, 3, 1)[:dim]
        lo = (0.24899, 0.19, -0.005)[:dim]
        hi = (0.6, 0.21, 0.005)[:dim]
        roles = [FACE_CLAMPED, FACE_INTERFACE, FACE_INTERFACE, FACE_INTERFACE, FACE_ZCLAMP, FACE_ZCLAMP]
    elif scenario == "PF":
        reps = (3, 18, 1)[:dim]
        lo = (flap_location - 0.05, 0.0, 0.0)[:dim]
        hi = (flap_location + 0.05, 1.0, 0.3)[:dim]
        roles = [FACE_INTERFACE, FACE_INTERFACE, FACE_CLAMPED, FACE_INTERFACE, FACE_ZCLAMP, FACE_ZCLAMP]
    else:
        raise ValueError(scenario)
    return make_desc(dim=dim, reps=reps, lo=lo, hi=hi, face_role=roles, **kw)


def material(dim, mu, nu, F):
    F3 = np.eye(3)
    F3[:dim, :dim] = np.asarray(F, dtype=np.float64)[:dim, :dim]
    F3 = np.ascontiguousarray(F3)
    tau = np.zeros((3, 3))
    Jc = np.zeros((3, 3, 3, 3))
    psi = lib().orc_material(dim, mu, nu, _dp(F3), _dp(tau), _dp(Jc))
    return psi, tau[:dim, :dim].copy(), Jc[:dim, :dim, :dim, :dim].copy()


def cell_tangent_residual(desc, verts, u, acc):
    dim, p = desc.dim, desc.degree
    dpc = dim * (p + 1) ** dim
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    Ke = np.zeros((dpc, dpc))
    re = np.zeros(dpc)
    lib().orc_cell_tangent_residual(C.byref(desc), _dp(verts), _dp(u), _dp(acc), _dp(Ke), _dp(re))
    return Ke, re


class Problem:
    """Nonlinear (neo-Hookean, Newmark) oracle problem."""

    def __init__(self, desc, perturb=None):
        self.desc = desc
        L = lib()
        if perturb is not None:
            perturb = np.ascontiguousarray(perturb, dtype=np.float64)
        self._perturb = perturb
        self.h = L.orc_create(C.byref(desc), _dp(perturb) if perturb is not None else None)
        self.n = L.orc_n_dofs(self.h)
        self.nnodes = L.orc_n_nodes(self.h)
        self.ncells = L.orc_n_cells(self.h)
        self.nnz = L.orc_nnz(self.h)
        self.dim = desc.dim

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def vec(self, which):
        return np.ctypeslib.as_array(lib().orc_vec(self.h, which), shape=(self.n,))

    @property
    def coords(self):
        return np.ctypeslib.as_array(lib().orc_node_coords(self.h), shape=(self.nnodes, self.dim))

    @property
    def constrained(self):
        return np.ctypeslib.as_array(lib().orc_constrained(self.h), shape=(self.n,)).astype(bool)

    @property
    def interface_nodes(self):
        k = lib().orc_n_interface_nodes(self.h)
        if k == 0:  # an empty std::vector hands out a null pointer
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(lib().orc_interface_nodes(self.h), shape=(k,)).copy()

    def csr(self):
        import scipy.sparse as sp
        L = lib()
        rp = np.ctypeslib.as_array(L.orc_csr_rowptr(self.h), shape=(self.n + 1,))
        col = np.ctypeslib.as_array(L.orc_csr_col(self.h), shape=(self.nnz,))
        val = np.ctypeslib.as_array(L.orc_csr_val(self.h), shape=(self.nnz,))
        return sp.csr_matrix((val.copy(), col.copy(), rp.copy()), shape=(self.n, self.n))

    def update_acceleration(self):
        lib().orc_update_acceleration(self.h)

    def assemble(self):
        lib().orc_assemble(self.h)

    def residual_norm(self):
        return lib().orc_residual_norm(self.h)

    def solve_linear(self, solver=SOLVER_CG_SSOR, tol_lin=1e-6, max_it_mult=1.0):
        its, res = C.c_int(0), C.c_double(0)
        rc = lib().orc_solve_linear(self.h, solver, tol_lin, max_it_mult, C.byref(its), C.byref(res))
        return rc, its.value, res.value

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        lib().orc_spmv(self.h, _dp(x), _dp(y))
        return y

    def newmark_step(self, solver=SOLVER_CG_SSOR, tol_lin=1e-6, max_it_mult=1.0, max_it_nr=10, tol_f=1e-9,
                     tol_u=1e-6):
        info = StepInfo()
        rc = lib().orc_newmark_step(self.h, solver, tol_lin, max_it_mult, max_it_nr, tol_f, tol_u, C.byref(info))
        return rc, info

    def set_interface_traction(self, t):
        """t: (dim,) constant or (n_iface, dim) per interface node (ascending node order)."""
        s = self.vec(V_STRESS)
        s[:] = 0.0
        nodes = self.interface_nodes
        t = np.broadcast_to(np.asarray(t, dtype=np.float64), (len(nodes), self.dim))
        for c in range(self.dim):
            s[nodes * self.dim + c] = t[:, c]


class LinearProblem:
    def __init__(self, desc):
        self.desc = desc
        L = lib()
        self.h = L.orc_linear_create(C.byref(desc))
        self.n = L.orc_linear_n_dofs(self.h)
        self.nnz = L.orc_linear_nnz(self.h)
        self.dim = desc.dim
        self.nnodes = self.n // self.dim

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_linear_destroy(self.h)
            self.h = None

    def vec(self, which):
        return np.ctypeslib.as_array(lib().orc_linear_vec(self.h, which), shape=(self.n,))

    def matrix(self, which):
        import scipy.sparse as sp
        L = lib()
        rp = np.ctypeslib.as_array(L.orc_linear_rowptr(self.h), shape=(self.n + 1,))
        col = np.ctypeslib.as_array(L.orc_linear_col(self.h), shape=(self.nnz,))
        val = np.ctypeslib.as_array(L.orc_linear_matrix(self.h, which), shape=(self.nnz,))
        return sp.csr_matrix((val.copy(), col.copy(), rp.copy()), shape=(self.n, self.n))

    @property
    def coords(self):
        return np.ctypeslib.as_array(lib().orc_linear_node_coords(self.h), shape=(self.nnodes, self.dim))

    @property
    def constrained(self):
        return np.ctypeslib.as_array(lib().orc_linear_constrained(self.h), shape=(self.n,)).astype(bool)

    @property
    def interface_nodes(self):
        k = lib().orc_linear_n_interface_nodes(self.h)
        if k == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(lib().orc_linear_interface_nodes(self.h), shape=(k,)).copy()

    def step(self, solver=SOLVER_DIRECT, data_consistent=True, abs_tol=None):
        its, res = C.c_int(0), C.c_double(0)
        if abs_tol is None:
            rc = lib().orc_linear_step(self.h, solver, int(data_consistent), C.byref(its), C.byref(res))
        else:
            rc = lib().orc_linear_step_tol(self.h, solver, int(data_consistent), abs_tol, C.byref(its), C.byref(res))
        return rc, its.value, res.value
