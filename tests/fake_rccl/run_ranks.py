#!/usr/bin/env python3
"""Driver of tests/test_gpu_fake_rccl.py: N rank THREADS in one process go through the library's RCCL branch (one
slab per rank, `mi_comm_desc` with a unique id) against the RCCL test double, and are compared with the undecomposed
and the emulated-slab runs of the same problem.  Prints one JSON line."""
import importlib.util
import json
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load():
    spec = importlib.util.spec_from_file_location("dealii_adapter_amd_fake", os.path.join(ROOT, "dealii-adapter_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dealii_adapter_amd_fake"] = mod
    spec.loader.exec_module(mod)
    mod.LIB_PATH = os.path.join(HERE, "libmi_elasticity_fakerccl.so")  # same objects, RCCL replaced by the test double
    return mod


def linear_scenario(G, M):
    """the linear theta-model: per-slab host assembly, both products and the PCG across the ranks"""
    import ctypes as C
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    assert L.mi_linear_setup(G.h, 0.6) == 0, L.mi_last_error(G.h)
    rng = np.random.default_rng(21)
    ids, _ = G.interface()
    out = {}
    for step in range(3):
        G.set_interface_traction(100.0 * rng.standard_normal((len(ids), G.dim)))
        its, res = C.c_int(0), C.c_double(0)
        rc = L.mi_linear_step(G.h, int(step != 1), 1e-12, G.n * 4, C.byref(its), C.byref(res))
        assert rc == 0, L.mi_last_error(G.h)
    out["d"] = G.get(0)
    out["v"] = G.get(2)
    return out


def scenario(G, M, kw, log):
    """the same API sequence on every rank (all arrays global)"""
    rng = np.random.default_rng(5)
    ids, _ = G.interface()
    out = {}
    G.set_tuning("halo_overlap", kw.get("overlap", 1))
    if kw.get("fine"):  # no assembled fine tangent: every product of the level on mf_spmv, the CG's around its halo exchange
        G.set_tuning("fine_level", 1)
    if kw.get("dist_nodes", -1) >= 0:  # shape of the hierarchy: before "precond" 1 builds it
        G.set_tuning("mg_dist_nodes", kw["dist_nodes"])
    for precond in (0, 1):
        G.set_tuning("precond", precond)
        if precond == 1 and kw.get("ebe"):
            # the smoother on the element tangents (as on big meshes): unfused smoother, every slab multiplies with all its
            # local cells after the halo exchange
            # (ebe = 2: matrix-free from the quadrature-point records, one launch + the gather that applies the step)
            G.set_tuning("element_tangents", int(kw["ebe"]))
            G.set_tuning("mg_fuse", 0)
            assert G.get_tuning("smoother_operator_active") == int(kw["ebe"])
        its = []
        for step in range(2):
            G.set_interface_traction((0.0, -1.5e3 * (step + 1), 0.0)[:G.dim])
            rc, info = G.newmark_step(tol_lin=1e-10, max_it_mult=2.0)
            assert rc == 0 and info.converged == 1, (rc, precond, step)
            its.append(int(info.lin_its_total))
        if precond == 1:
            out["mg_dist_levels"] = G.get_tuning("mg_distributed_levels")
        out["its%d" % precond] = its
        out["u%d" % precond] = G.get(M.V_U)
        out["if%d" % precond] = G.get_interface_displacement()
    x = np.random.default_rng(9).standard_normal(G.n)
    G.update_acceleration()
    out["rn"] = G.assemble()
    out["Kx"] = G.spmv(x)
    G.state_save()
    G.set(M.V_U, rng.standard_normal(G.n))
    G.state_restore()
    out["restored"] = G.get(M.V_U)
    lin = linear_scenario(G, M)
    out["lin_d"], out["lin_v"] = lin["d"], lin["v"]
    # rank 0's values to every rank (what carries the ONE preCICE-facing process's answers to the others): each rank offers
    # different numbers, longer than the interface scratch (several chunks), with a -0.0 and a huge value among them
    rank = kw.get("rank", 0)
    mine = (rank + 1.0) * np.concatenate([[-0.0, 1e300, -3.5], np.arange(3 * G.dim * len(ids) + 11, dtype=np.float64)])
    out["bcast"] = G.comm_broadcast(mine)
    return out


def main():
    world = int(sys.argv[1])
    dim, p = int(sys.argv[2]), int(sys.argv[3])
    reps = tuple(int(v) for v in sys.argv[4].split(","))
    overlap = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    ebe = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    dist_nodes = int(sys.argv[7]) if len(sys.argv) > 7 else -1  # tuning "mg_dist_nodes" (-1: the library's default)
    fine = int(sys.argv[8]) if len(sys.argv) > 8 else 0           # tuning "fine_level" (1: the fine level matrix-free, 3D Q2)
    M = load()
    hi = tuple(0.1 * r for r in reps)
    roles = [1, 7, 7, 7, 8, 7]
    common = dict(dim=dim, degree=p, reps=reps, hi=hi, face_role=roles)
    uid = M.comm_unique_id()
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            G = M.Context(rank=r, world=world, unique_id=uid, **common)
            results[r] = scenario(G, M, dict(overlap=overlap, ebe=ebe, rank=r, dist_nodes=dist_nodes, fine=fine), None)
            G.close()
        except BaseException as e:  # noqa: BLE001 -- reported to the parent test
            errors.append("rank %d: %r" % (r, e))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if errors or any(t.is_alive() for t in threads):
        print(json.dumps({"ok": False, "errors": errors, "hung": [t.is_alive() for t in threads]}), flush=True)
        os._exit(1)
    single = scenario(M.Context(**common), M, dict(ebe=ebe, dist_nodes=dist_nodes, fine=fine), None)
    emu = scenario(M.Context(slabs=world, **common), M, dict(ebe=ebe, dist_nodes=dist_nodes, fine=fine), None)

    def rel(a, b):
        return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))

    rep = {"ok": True, "world": world, "its_ranks": [results[r]["its0"] + results[r]["its1"] for r in range(world)],
           "its_single": single["its0"] + single["its1"], "its_emulated": emu["its0"] + emu["its1"]}
    ref = np.concatenate([[0.0, 1e300, -3.5], np.arange(results[0]["bcast"].size - 3, dtype=np.float64)])
    # bit patterns, not values: the -0.0 that rank 0 sent arrives as +0.0 on every rank of a team of several (x + 0 + ... + 0
    # in the all-reduce; documented at mi_comm_broadcast) and stays -0.0 where the call is a no-op (one rank)
    ref_bits = ref.view(np.uint64).copy()
    if world == 1:
        ref_bits[0] = np.float64(-0.0).view(np.uint64)
    rep["broadcast_ok"] = bool(all(np.array_equal(np.asarray(results[r]["bcast"], dtype=np.float64).view(np.uint64), ref_bits)
                                   for r in range(world)))
    keys = ("u0", "u1", "if0", "if1", "Kx", "restored", "lin_d", "lin_v")
    rep["rank_spread"] = max(rel(results[r][k], results[0][k]) for r in range(1, world) for k in keys) if world > 1 else 0.0
    rep["vs_single"] = {k: rel(results[0][k], single[k]) for k in keys}
    rep["vs_emulated"] = {k: rel(results[0][k], emu[k]) for k in keys}
    rep["rn"] = [results[r]["rn"] for r in range(world)] + [single["rn"], emu["rn"]]
    rep["mg_dist_levels"] = [results[0]["mg_dist_levels"], emu["mg_dist_levels"], single["mg_dist_levels"]]
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
