"""ctypes plumbing over libmi_elasticity.so (C-ABI: include/mi_elasticity.h).

This module is test/bench plumbing only: the product's host side is C++ (dealii-adapter_amd/host/), as in
the reference.  There is no CPU fallback: if the HIP library is missing or no device is visible, loading or
context creation raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("MI_LIB") or os.path.join(PKG_DIR, "libmi_elasticity.so")  # MI_LIB: A/B of two builds in one job (tools)
HEADER = os.path.join(ROOT, "include", "mi_elasticity.h")

MI_OK, MI_EINVAL, MI_EHIP, MI_ENOCONV_LIN, MI_ENOCONV_NR, MI_ECOMM = 0, -1, -2, -3, -4, -5
FACE_FREE, FACE_CLAMPED, FACE_INTERFACE, FACE_ZCLAMP = 0, 1, 7, 8
(V_U, V_U_OLD, V_V, V_V_OLD, V_A, V_A_OLD, V_STRESS, V_DELTA, V_NEWTON, V_RHS) = range(10)
(T_ASSEMBLE_CELLS, T_ASSEMBLE_TOTAL, T_SPMV, T_CG_VECTOR, T_CG_TOTAL, T_NEWMARK, T_STEP, T_ASSEMBLE_DIAG, T_ASSEMBLE_RESIDUAL,
 T_SPMV_PRECOND, T_EBE_LAUNCH, T_COUNT) = range(12)
TIMING_NAMES = ["assemble_cells", "assemble_total", "spmv", "cg_vector", "cg_total", "newmark", "step", "assemble_diag",
                "assemble_residual", "spmv_precond", "ebe_launch"]


class MeshDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("degree", C.c_int32), ("reps", C.c_int32 * 3), ("lo", C.c_double * 3),
                ("hi", C.c_double * 3), ("face_role", C.c_int32 * 6), ("vertex_perturbation", C.POINTER(C.c_double))]


class MaterialDesc(C.Structure):
    _fields_ = [("mu", C.c_double), ("nu", C.c_double), ("rho", C.c_double), ("body_force", C.c_double * 3)]


class NewmarkDesc(C.Structure):
    _fields_ = [("beta", C.c_double), ("gamma", C.c_double), ("delta_t", C.c_double)]


class CommDesc(C.Structure):
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("nccl_unique_id", C.c_void_p), ("cut_axis", C.c_int32)]


class PartitionInfo(C.Structure):
    _fields_ = [("z0", C.c_int32), ("z1", C.c_int32), ("local_layers", C.c_int32), ("plane_nodes", C.c_int64),
                ("node_offset", C.c_int64), ("nnodes_global", C.c_int64), ("nnodes_local", C.c_int64),
                ("own_begin", C.c_int64), ("own_end", C.c_int64), ("up_send", C.c_int64), ("up_send_n", C.c_int64),
                ("up_recv", C.c_int64), ("up_recv_n", C.c_int64), ("down_send", C.c_int64), ("down_send_n", C.c_int64),
                ("down_recv", C.c_int64), ("down_recv_n", C.c_int64), ("local_reps", C.c_int32 * 3),
                ("local_lo", C.c_double * 3), ("local_hi", C.c_double * 3), ("local_face_role", C.c_int32 * 6)]


class SolverDesc(C.Structure):
    _fields_ = [("tol_lin", C.c_double), ("max_iterations_lin", C.c_double), ("max_iterations_NR", C.c_int32),
                ("tol_f", C.c_double), ("tol_u", C.c_double)]


class StepInfo(C.Structure):
    _fields_ = [("newton_iterations", C.c_int32), ("assemblies", C.c_int32), ("lin_its_total", C.c_int32),
                ("converged", C.c_int32), ("res_norm", C.c_double), ("res_abs", C.c_double),
                ("upd_norm", C.c_double), ("upd_abs", C.c_double), ("lin_its", C.c_int32 * 16),
                ("lin_res", C.c_double * 16)]


class Timings(C.Structure):
    _fields_ = [("ms", C.c_double * T_COUNT), ("count", C.c_int64 * T_COUNT)]


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi_elasticity error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)"""
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"])
    subprocess.check_call(["make", "-C", PKG_DIR, "-j4", "all"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `make -C dealii-adapter_amd` (no CPU fallback exists)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        dp, i32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
        L.mi_ctx_create.restype = C.c_int
        L.mi_ctx_create.argtypes = [C.POINTER(MeshDesc), C.POINTER(MaterialDesc), C.POINTER(NewmarkDesc), C.c_int,
                                    C.POINTER(CommDesc), C.POINTER(vp)]
        L.mi_ctx_destroy.restype = None
        L.mi_ctx_destroy.argtypes = [vp]
        L.mi_last_error.restype = C.c_char_p
        L.mi_last_error.argtypes = [vp]
        for f in ("mi_n_dofs", "mi_n_nodes", "mi_n_cells", "mi_nnz"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [vp]
        for f in ("mi_n_colours", "mi_n_interface_nodes", "mi_newton_begin_step", "mi_update_acceleration",
                  "mi_newmark_finish_step", "mi_state_save", "mi_state_restore", "mi_reset_timings"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp]
        L.mi_get_node_coords.argtypes = [vp, dp]
        L.mi_get_constrained.argtypes = [vp, C.POINTER(C.c_uint8)]
        L.mi_get_interface_nodes.argtypes = [vp, i32p, dp]
        L.mi_set_interface_traction.argtypes = [vp, C.c_int, dp]
        L.mi_get_interface_displacement.argtypes = [vp, C.c_int, dp]
        L.mi_assemble.argtypes = [vp, dp]
        L.mi_assemble_residual.argtypes = [vp, dp]
        L.mi_assemble_residual.restype = C.c_int
        L.mi_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi_comm_info.restype = C.c_int
        L.mi_comm_broadcast.argtypes = [vp, C.POINTER(C.c_double), C.c_int32]
        L.mi_comm_broadcast.restype = C.c_int
        L.mi_cg_solve.argtypes = [vp, C.c_double, C.c_int64, C.POINTER(C.c_int), dp]
        L.mi_apply_newton_update.argtypes = [vp, dp]
        L.mi_direct_solve.argtypes = [vp, dp]
        L.mi_direct_solve.restype = C.c_int
        L.mi_newmark_step.argtypes = [vp, C.POINTER(SolverDesc), C.POINTER(StepInfo)]
        L.mi_vec_get.argtypes = [vp, C.c_int, dp, C.c_int64]
        L.mi_vec_set.argtypes = [vp, C.c_int, dp, C.c_int64]
        L.mi_matrix_get_csr.argtypes = [vp, C.POINTER(C.c_int64), i32p, dp]
        L.mi_spmv.argtypes = [vp, dp, dp]
        L.mi_get_diagonal_blocks.argtypes = [vp, dp]
        L.mi_get_diagonal_blocks.restype = C.c_int
        L.mi_set_profiling.argtypes = [vp, C.c_int]
        L.mi_get_timings.argtypes = [vp, C.POINTER(Timings)]
        L.mi_partition_describe.restype = C.c_int
        L.mi_partition_describe.argtypes = [C.POINTER(MeshDesc), C.c_int, C.c_int, C.POINTER(PartitionInfo)]
        L.mi_partition_spmv_rows.restype = C.c_int
        L.mi_partition_spmv_rows.argtypes = [C.POINTER(MeshDesc), C.c_int, C.c_int, C.POINTER(C.c_int64),
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64]
        L.mi_comm_unique_id.restype = C.c_int
        L.mi_comm_unique_id.argtypes = [C.c_void_p]
        L.mi_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
        L.mi_set_tuning.restype = C.c_int
        L.mi_get_tuning.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.mi_get_tuning.restype = C.c_int
        L.mi_bench_spmv.argtypes = [vp, C.c_int, dp]
        L.mi_bench_assemble.argtypes = [vp, C.c_int, dp]
        for f in ("mi_get_node_coords", "mi_get_constrained", "mi_get_interface_nodes", "mi_set_interface_traction",
                  "mi_get_interface_displacement", "mi_assemble", "mi_cg_solve", "mi_apply_newton_update",
                  "mi_newmark_step", "mi_vec_get", "mi_vec_set", "mi_matrix_get_csr", "mi_spmv", "mi_set_profiling",
                  "mi_get_timings", "mi_bench_spmv", "mi_bench_assemble"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def declared_symbols():
    """entry points declared in include/mi_elasticity.h"""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt)))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def mesh_desc(dim, degree, reps, lo, hi, face_role):
    md = MeshDesc()
    md.dim, md.degree = dim, degree
    reps = tuple(reps) + (1,) * (3 - len(reps))
    lo = tuple(lo) + (0.0,) * (3 - len(lo))
    hi = tuple(hi) + (0.0,) * (3 - len(hi))
    for i in range(3):
        md.reps[i], md.lo[i], md.hi[i] = reps[i], lo[i], hi[i]
    for i in range(6):
        md.face_role[i] = face_role[i]
    return md


def partition_describe(md, rank, size):
    """host-only slab description (works without a GPU)"""
    info = PartitionInfo()
    rc = lib().mi_partition_describe(C.byref(md), rank, size, C.byref(info))
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error(None).decode())
    return info


def partition_spmv_rows(md, rank, size):
    """host-only: (rows, n_interior_slots) -- SpMV row order of a slab, interior rows first (-1 = padding)"""
    ns, ni = C.c_int64(), C.c_int64()
    rc = lib().mi_partition_spmv_rows(C.byref(md), rank, size, C.byref(ns), C.byref(ni), None, 0)
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error(None).decode())
    rows = np.zeros(ns.value * 64, dtype=np.int32)
    rc = lib().mi_partition_spmv_rows(C.byref(md), rank, size, C.byref(ns), C.byref(ni),
                                      rows.ctypes.data_as(C.POINTER(C.c_int32)), rows.size)
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error(None).decode())
    return rows, ni.value * 64


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().mi_comm_unique_id(buf)
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error(None).decode())
    return buf.raw


class Context:
    """one device context = one (sub)domain of the structural problem on one GPU"""

    def __init__(self, dim=3, degree=2, reps=(4, 4, 4), lo=(0, 0, 0), hi=(1, 1, 1), face_role=None, mu=0.5e6, nu=0.4,
                 rho=1000.0, body_force=(0, 0, 0), beta=0.25, gamma=0.5, delta_t=0.005, device=0, perturb=None,
                 slabs=1, rank=None, world=1, unique_id=None, cut_axis=0):
        """slabs > 1: that many slabs inside this process (test mode); rank/world/unique_id: one slab per process
        over RCCL; cut_axis: 0 automatic (direction with most cell layers, ties: the last), 1 / 2 / 3 = x / y / z"""
        L = lib()
        md, mat, nm = MeshDesc(), MaterialDesc(), NewmarkDesc()
        md.dim, md.degree = dim, degree
        reps = tuple(reps) + (1,) * (3 - len(reps))
        lo = tuple(lo) + (0.0,) * (3 - len(lo))
        hi = tuple(hi) + (0.0,) * (3 - len(hi))
        if face_role is None:
            face_role = [FACE_CLAMPED] + [FACE_INTERFACE] * 5
        for i in range(3):
            md.reps[i], md.lo[i], md.hi[i], mat.body_force[i] = reps[i], lo[i], hi[i], body_force[i]
        for i in range(6):
            md.face_role[i] = face_role[i]
        self._perturb = None
        if perturb is not None:
            self._perturb = np.ascontiguousarray(perturb, dtype=np.float64)
            md.vertex_perturbation = _dp(self._perturb)
        mat.mu, mat.nu, mat.rho = mu, nu, rho
        nm.beta, nm.gamma, nm.delta_t = beta, gamma, delta_t
        self.h = C.c_void_p()
        comm = None
        if slabs > 1:
            comm = CommDesc(-1, slabs, None, cut_axis)
        elif world > 1 or unique_id is not None:
            self._uid = C.create_string_buffer(unique_id, 128)
            comm = CommDesc(rank or 0, world, C.cast(self._uid, C.c_void_p), cut_axis)
        rc = L.mi_ctx_create(C.byref(md), C.byref(mat), C.byref(nm), device, C.byref(comm) if comm else None,
                             C.byref(self.h))
        if rc != MI_OK:
            self.h = None
            raise MiError(rc, L.mi_last_error(None).decode())
        self.dim = dim
        self.n = L.mi_n_dofs(self.h)
        self.nnodes = L.mi_n_nodes(self.h)
        self.ncells = L.mi_n_cells(self.h)
        self.nnz = L.mi_nnz(self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().mi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != MI_OK:
            raise MiError(rc, lib().mi_last_error(self.h).decode())

    @property
    def coords(self):
        x = np.zeros((self.nnodes, self.dim))
        self._chk(lib().mi_get_node_coords(self.h, _dp(x)))
        return x

    @property
    def constrained(self):
        f = np.zeros(self.n, dtype=np.uint8)
        self._chk(lib().mi_get_constrained(self.h, f.ctypes.data_as(C.POINTER(C.c_uint8))))
        return f.astype(bool)

    def interface(self):
        k = lib().mi_n_interface_nodes(self.h)
        ids = np.zeros(k, dtype=np.int32)
        xyz = np.zeros((k, self.dim))
        self._chk(lib().mi_get_interface_nodes(self.h, ids.ctypes.data_as(C.POINTER(C.c_int32)), _dp(xyz)))
        return ids, xyz

    def set_interface_traction(self, t):
        k = lib().mi_n_interface_nodes(self.h)
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (k, self.dim)))
        self._chk(lib().mi_set_interface_traction(self.h, k, _dp(t)))

    def get_interface_displacement(self):
        k = lib().mi_n_interface_nodes(self.h)
        out = np.zeros((k, self.dim))
        self._chk(lib().mi_get_interface_displacement(self.h, k, _dp(out)))
        return out

    def get(self, which):
        v = np.zeros(self.n)
        self._chk(lib().mi_vec_get(self.h, which, _dp(v), self.n))
        return v

    def set(self, which, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self._chk(lib().mi_vec_set(self.h, which, _dp(v), self.n))

    def newton_begin_step(self):
        self._chk(lib().mi_newton_begin_step(self.h))

    def update_acceleration(self):
        self._chk(lib().mi_update_acceleration(self.h))

    def assemble(self):
        r = C.c_double(0)
        self._chk(lib().mi_assemble(self.h, C.byref(r)))
        return r.value

    def assemble_residual(self):
        r = C.c_double(0)
        self._chk(lib().mi_assemble_residual(self.h, C.byref(r)))
        return r.value

    def comm_info(self):
        """(slabs of the decomposition, ncclCommCount of the RCCL communicator or 0)"""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(lib().mi_comm_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def cg_solve(self, rel_tol=1e-6, max_it=None):
        its, res = C.c_int(0), C.c_double(0)
        rc = lib().mi_cg_solve(self.h, rel_tol, self.n if max_it is None else max_it, C.byref(its), C.byref(res))
        if rc not in (MI_OK, MI_ENOCONV_LIN):
            self._chk(rc)
        return rc, its.value, res.value

    def direct_solve(self):
        """banded Cholesky of the current tangent + substitution (mi_direct_solve); returns the status code"""
        r = C.c_double(0)
        rc = lib().mi_direct_solve(self.h, C.byref(r))
        if rc not in (MI_OK, MI_EINVAL, MI_ENOCONV_LIN):
            self._chk(rc)
        return rc

    def apply_newton_update(self):
        r = C.c_double(0)
        self._chk(lib().mi_apply_newton_update(self.h, C.byref(r)))
        return r.value

    def newmark_finish_step(self):
        self._chk(lib().mi_newmark_finish_step(self.h))

    def newmark_step(self, tol_lin=1e-6, max_it_mult=1.0, max_it_nr=10, tol_f=1e-9, tol_u=1e-6, check=True):
        s, info = SolverDesc(), StepInfo()
        s.tol_lin, s.max_iterations_lin, s.max_iterations_NR, s.tol_f, s.tol_u = (tol_lin, max_it_mult, max_it_nr,
                                                                                  tol_f, tol_u)
        rc = lib().mi_newmark_step(self.h, C.byref(s), C.byref(info))
        if check:
            self._chk(rc)
        return rc, info

    def comm_broadcast(self, values):
        """values of rank 0 to every rank (collective; returns the received array)"""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(lib().mi_comm_broadcast(self.h, _dp(v), v.size))
        return v

    def state_save(self):
        self._chk(lib().mi_state_save(self.h))

    def state_restore(self):
        self._chk(lib().mi_state_restore(self.h))

    def csr(self):
        import scipy.sparse as sp
        rp = np.zeros(self.n + 1, dtype=np.int64)
        col = np.zeros(self.nnz, dtype=np.int32)
        val = np.zeros(self.nnz)
        self._chk(lib().mi_matrix_get_csr(self.h, rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                          col.ctypes.data_as(C.POINTER(C.c_int32)), _dp(val)))
        return sp.csr_matrix((val, col, rp), shape=(self.n, self.n))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self._chk(lib().mi_spmv(self.h, _dp(x), _dp(y)))
        return y

    def diagonal_blocks(self):
        """[n_nodes, dim, dim]: the diagonal block of every node of the current tangent"""
        b = np.zeros((self.nnodes, self.dim, self.dim))
        self._chk(lib().mi_get_diagonal_blocks(self.h, _dp(b)))
        return b

    def set_profiling(self, on=True):
        self._chk(lib().mi_set_profiling(self.h, int(on)))

    def set_tuning(self, key, value):
        self._chk(lib().mi_set_tuning(self.h, key.encode(), int(value)))

    def get_tuning(self, key):
        v = C.c_int(0)
        self._chk(lib().mi_get_tuning(self.h, key.encode(), C.byref(v)))
        return v.value

    def reset_timings(self):
        self._chk(lib().mi_reset_timings(self.h))

    def timings(self):
        t = Timings()
        self._chk(lib().mi_get_timings(self.h, C.byref(t)))
        return {TIMING_NAMES[i]: (t.ms[i], t.count[i]) for i in range(T_COUNT)}

    def bench_spmv(self, reps=20):
        ms = C.c_double(0)
        self._chk(lib().mi_bench_spmv(self.h, reps, C.byref(ms)))
        return ms.value

    def bench_assemble(self, reps=3):
        ms = C.c_double(0)
        self._chk(lib().mi_bench_assemble(self.h, reps, C.byref(ms)))
        return ms.value
