"""CPU tests of the multi-GPU path's host logic, including a world-size-2 run over gloo.

The slab description comes from the product library (mi_partition_describe, pure host code).  Two processes
then emulate what two GPUs do: each builds the local box of its slab (own layers + one ghost layer) with the CPU
oracle as the local assembler, assembles redundantly, multiplies its OWNED rows after exchanging the halo planes
over gloo, and runs a distributed Jacobi-PCG with all-reduced scalars.  Rank 0 checks against the undecomposed
oracle.  This pins the ownership rule, the sufficiency of one ghost layer and the halo ranges the RCCL path uses.
"""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

M = load_pkg()

ROLES = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_INTERFACE]


def _md(dim, p, reps, hi):
    return M.mesh_desc(dim, p, reps, (0.0,) * dim, hi, ROLES)


@pytest.mark.parametrize("dim,p,reps,size", [(3, 2, (3, 2, 7), 3), (3, 1, (4, 4, 59), 8), (2, 3, (5, 9), 4)])
def test_slabs_tile_the_mesh(dim, p, reps, size):
    md = _md(dim, p, reps, tuple(0.1 * r for r in reps))
    infos = [M.partition_describe(md, r, size) for r in range(size)]
    plane = infos[0].plane_nodes
    assert infos[0].nnodes_global == plane * (p * reps[-1] + 1)
    covered = 0
    for r, s in enumerate(infos):
        assert s.z1 > s.z0 and (r == 0 or s.z0 == infos[r - 1].z1)
        assert s.local_layers == s.z1 - s.z0 + (1 if r < size - 1 else 0)
        assert s.node_offset == plane * p * s.z0
        g0, g1 = s.node_offset + s.own_begin, s.node_offset + s.own_end
        assert g0 == covered  # owned global ranges are contiguous and disjoint
        covered = g1
        if r < size - 1:  # what I send up is what the next rank receives from below, and vice versa
            nxt = infos[r + 1]
            assert s.up_send_n == nxt.down_recv_n == plane and s.up_recv_n == nxt.down_send_n == p * plane
            assert s.node_offset + s.up_send == nxt.node_offset + nxt.down_recv
            assert s.node_offset + s.up_recv == nxt.node_offset + nxt.down_send
            # interior cut, unless the ghost layer is the topmost layer of the mesh (then the real role applies)
            assert s.local_face_role[2 * dim - 1] == (ROLES[2 * dim - 1] if s.z1 + 1 == reps[-1] else 0)
        else:
            assert s.up_send_n == 0 and s.local_face_role[2 * dim - 1] == ROLES[2 * dim - 1]
        assert s.local_face_role[2 * dim - 2] == (ROLES[2 * dim - 2] if r == 0 else 0)
    assert covered == infos[0].nnodes_global
    assert infos[-1].z1 == reps[-1]
    with pytest.raises(M.MiError):
        M.partition_describe(md, 0, reps[-1] + 1)


def _worker(rank, world, port, dim, p, reps, out_q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hi = tuple(0.1 * r for r in reps)
        md = _md(dim, p, reps, hi)
        s = M.partition_describe(md, rank, world)
        D = dim
        loc = O.make_desc(dim=dim, degree=p, reps=tuple(s.local_reps)[:dim], lo=tuple(s.local_lo)[:dim],
                          hi=tuple(s.local_hi)[:dim], face_role=list(s.local_face_role))
        P = O.Problem(loc)
        assert P.nnodes == s.nnodes_local
        own = slice(s.own_begin * D, s.own_end * D)
        glob = slice(s.node_offset * D, (s.node_offset + s.nnodes_local) * D)
        # identical global state on every rank (seeded), each takes its local part incl. ghosts
        rng = np.random.default_rng(7)
        ng = s.nnodes_global * D
        Pg = O.Problem(O.make_desc(dim=dim, degree=p, reps=reps, hi=hi, face_role=ROLES)) if rank == 0 else None
        state = {k: rng.standard_normal(ng) * sc for k, sc in ((O.V_U, 1e-4), (O.V_V_OLD, 0.1), (O.V_A_OLD, 1.0))}
        for k, v in state.items():
            P.vec(k)[:] = v[glob]
            if Pg:
                Pg.vec(k)[:] = v
        for prob in (P, Pg):
            if prob:
                prob.vec(O.V_U)[prob.constrained] = 0
                prob.set_interface_traction((0.0, -2e3, 0.0)[:dim])
                prob.update_acceleration()
                prob.assemble()  # ghost cells are assembled redundantly, no matrix communication
        A = P.csr()

        def halo(v):  # same ranges as mi_ctx.cpp team_halo
            reqs = []
            if s.up_send_n:
                up = torch.from_numpy(np.ascontiguousarray(v[s.up_send * D:(s.up_send + s.up_send_n) * D]))
                rb = torch.zeros(s.up_recv_n * D, dtype=torch.float64)
                reqs += [dist.isend(up, rank + 1), dist.irecv(rb, rank + 1)]
            if s.down_send_n:
                dn = torch.from_numpy(np.ascontiguousarray(v[s.down_send * D:(s.down_send + s.down_send_n) * D]))
                rd = torch.zeros(s.down_recv_n * D, dtype=torch.float64)
                reqs += [dist.isend(dn, rank - 1), dist.irecv(rd, rank - 1)]
            for r in reqs:
                r.wait()
            if s.up_send_n:
                v[s.up_recv * D:(s.up_recv + s.up_recv_n) * D] = rb.numpy()
            if s.down_send_n:
                v[s.down_recv * D:(s.down_recv + s.down_recv_n) * D] = rd.numpy()

        def allsum(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t[0])

        def gather_owned(v):
            full = torch.zeros(ng, dtype=torch.float64)
            full[(s.node_offset + s.own_begin) * D:(s.node_offset + s.own_end) * D] = torch.from_numpy(v[own].copy())
            dist.all_reduce(full)
            return full.numpy()

        # (1) right-hand side and SpMV on owned rows
        rhs_g = gather_owned(P.vec(O.V_RHS))
        x = np.random.default_rng(4321).standard_normal(ng)
        xl = x[glob].copy()
        halo(xl)  # no-op numerically (already consistent) but exercises the ranges
        yl = A @ xl
        y_g = gather_owned(yl)
        # (2) distributed Jacobi-PCG, x0 = 0
        b = P.vec(O.V_RHS).copy()
        dinv = 1.0 / A.diagonal()
        xs, r = np.zeros_like(b), b.copy()
        bnorm = np.sqrt(allsum(b[own] @ b[own]))
        pvec, rz_old, its = np.zeros_like(b), 0.0, 0
        while True:
            rr, rz = allsum(r[own] @ r[own]), allsum(r[own] @ (dinv[own] * r[own]))
            if np.sqrt(rr) <= 1e-10 * bnorm or its > 5000:
                break
            its += 1
            beta = 0.0 if its == 1 else rz / rz_old
            rz_old = rz
            pvec[own] = dinv[own] * r[own] + beta * pvec[own]
            halo(pvec)
            q = A @ pvec
            alpha = rz / allsum(pvec[own] @ q[own])
            xs[own] += alpha * pvec[own]
            r[own] -= alpha * q[own]
        x_g = gather_owned(xs)
        if rank == 0:
            errs = {
                "rhs": np.abs(rhs_g - Pg.vec(O.V_RHS)).max() / np.abs(Pg.vec(O.V_RHS)).max(),
                "spmv": np.abs(y_g - Pg.csr() @ x).max() / np.abs(Pg.csr() @ x).max(),
            }
            Pg.vec(O.V_NEWTON)[:] = 0
            rc, its_ref, _ = Pg.solve_linear(O.SOLVER_CG_JACOBI, tol_lin=1e-10, max_it_mult=2.0)
            ref = Pg.vec(O.V_NEWTON)
            errs["cg"] = np.abs(x_g - ref).max() / np.abs(ref).max()
            errs["its"] = abs(its - its_ref)
            out_q.put(errs)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("dim,p,reps", [(3, 2, (2, 2, 4)), (2, 2, (6, 7))])
def test_two_rank_gloo_decomposition(dim, p, reps):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dim, p, reps, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    errs = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert errs["rhs"] < 1e-12 and errs["spmv"] < 1e-12
    assert errs["cg"] < 1e-7 and errs["its"] <= 1


@pytest.mark.parametrize("dim,p,reps,size", [(3, 2, (3, 2, 7), 3), (3, 1, (3, 3, 9), 4), (2, 3, (5, 9), 4), (3, 2, (2, 2, 2), 2)])
def test_spmv_rows_interior_first(dim, p, reps, size):
    """the SpMV covers exactly the owned rows; the rows without ghost columns come first (they are computed while
    the halo exchange is in flight), the rows that read ghost planes last"""
    md = _md(dim, p, reps, tuple(0.1 * r for r in reps))
    for r in range(size):
        s = M.partition_describe(md, r, size)
        rows, n_int = M.partition_spmv_rows(md, r, size)
        real = rows[rows >= 0]
        assert np.array_equal(np.sort(real), np.arange(s.own_begin, s.own_end))  # every owned row once, no ghost row
        # a row couples to all nodes of its cells: it reads a ghost plane iff it lies in the cell layer above the
        # lower ghost plane (rank > 0) or on the top owned plane below the ghost layer (rank < size-1)
        plane = rows // s.plane_nodes
        lo_ghost = s.down_recv_n > 0
        hi_ghost = s.up_recv_n > 0
        own_planes = (s.own_begin // s.plane_nodes, s.own_end // s.plane_nodes)
        expect_bnd = np.zeros(rows.size, dtype=bool)
        if lo_ghost:
            expect_bnd |= (plane >= own_planes[0]) & (plane < own_planes[0] + p)
        if hi_ghost:
            expect_bnd |= plane == own_planes[1] - 1
        is_bnd = np.arange(rows.size) >= n_int
        ok = rows >= 0
        assert np.array_equal(is_bnd[ok], expect_bnd[ok])
        if size > 1 and s.z1 - s.z0 >= 3:
            assert 0 < n_int < rows.size
    rows, n_int = M.partition_spmv_rows(md, 0, 1)
    assert n_int == rows.size  # undecomposed: a single launch over all rows


def test_random_partitions_tile_and_order_rows():
    """40 seeded random (dim, degree, cells, ranks): owned ranges tile the global node range, halo ranges pair up,
    the SpMV rows are exactly the owned rows with the ghost-reading ones last"""
    rng = np.random.default_rng(77)
    for _ in range(40):
        dim = int(rng.integers(2, 4))
        p = int(rng.integers(1, 5 if dim == 2 else 3))
        reps = tuple(int(rng.integers(1, 7)) for _ in range(dim))
        size = int(rng.integers(1, reps[-1] + 1))
        md = _md(dim, p, reps, tuple(0.1 * r for r in reps))
        covered = 0
        infos = [M.partition_describe(md, r, size) for r in range(size)]
        for r, s in enumerate(infos):
            assert s.node_offset + s.own_begin == covered
            covered = s.node_offset + s.own_end
            if r + 1 < size:
                nxt = infos[r + 1]
                assert (s.up_send_n, s.up_recv_n) == (nxt.down_recv_n, nxt.down_send_n) == (s.plane_nodes, p * s.plane_nodes)
                assert s.node_offset + s.up_send == nxt.node_offset + nxt.down_recv
                assert s.node_offset + s.up_recv == nxt.node_offset + nxt.down_send
            rows, n_int = M.partition_spmv_rows(md, r, size)
            real = rows[rows >= 0]
            assert np.array_equal(np.sort(real), np.arange(s.own_begin, s.own_end)), (dim, p, reps, size, r)
            plane = rows // s.plane_nodes
            lo_plane, hi_plane = s.own_begin // s.plane_nodes, s.own_end // s.plane_nodes
            bnd = np.zeros(rows.size, dtype=bool)
            if s.down_recv_n > 0:
                bnd |= (plane >= lo_plane) & (plane < lo_plane + p)
            if s.up_recv_n > 0:
                bnd |= plane == hi_plane - 1
            ok = rows >= 0
            assert np.array_equal((np.arange(rows.size) >= n_int)[ok], bnd[ok]), (dim, p, reps, size, r)
        assert covered == infos[0].nnodes_global
