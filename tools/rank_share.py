"""What ONE rank of an N-slab run of the headline block computes, measured directly: the 59 x 59 x ceil(59/N + 1) cell
slab (own layers + the ghost layer) as a problem of its own on the one GPU -- same kernels at the launch sizes a rank
really has (the serialised emulation of `bench.py --slabs N` adds up eight ranks AND eight copies of the replicated
levels; this is one rank, with a coarse hierarchy of its own slab standing in for the replicated one).  No
communication: the latency of halos and all-reduces comes on top.
  python tools/rank_share.py [n = 59] [slab counts = 1,2,4,8]"""
import importlib, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
M = importlib.import_module("dealii-adapter_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
for fine, N in [(f, c) for f in (0, 1) for c in counts]:  # (round 6: also with the fine level matrix-free, "fine_level" 1)
    nz = n if N == 1 else int(math.ceil(n / N)) + 1
    G = M.Context(dim=3, degree=2, reps=(n, n, nz), hi=(1.0, 1.0, nz / n))
    G.set_tuning("cg_warm_start", 2)
    G.set_tuning("precond", 1)
    if fine:
        G.set_tuning("fine_level", 1)
        G.set_tuning("mf_diag_lag", 1)
    steps, warm = 8, 2
    its = 0
    for k in range(warm + steps):
        if k == warm:
            G.sync() if hasattr(G, "sync") else None
            t0 = time.perf_counter()
        G.set_interface_traction((0.0, -2e3 * min(1.0, (k + 1) / 10.0), 0.0))
        rc, info = G.newmark_step(tol_lin=1e-6)
        assert rc == 0
        if k >= warm:
            its += info.lin_its_total
    G.get_interface_displacement()  # (synchronises)
    dt = (time.perf_counter() - t0) / steps
    print(("matrix-free fine level, " if fine else "") + "N = %d: slab of %d x %d x %d cells (%d DoFs): %.2f ms per step, %.1f CG iterations per step" % (
        N, n, n, nz, G.n, 1e3 * dt, its / steps), flush=True)
    G.close()
