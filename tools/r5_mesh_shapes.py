"""Round 5 check of the multigrid hierarchy's new end (4^dim cells, dense inverse; thin directions keep coarsening) over mesh
shapes: two Newmark steps with the default solver, iteration counts with MI_MG_COARSEST=2 (rounds 2-4) for comparison in a second
process.  python tools/r5_mesh_shapes.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import _pkg
M = _pkg()
for dim, p, reps in ((3, 2, (59, 59, 9)), (3, 2, (100, 20, 6)), (3, 2, (30, 30, 120)), (3, 2, (24, 24, 24)), (3, 1, (96, 96, 20)),
                     (3, 1, (64, 64, 64)), (2, 2, (600, 90)), (2, 3, (300, 300)), (3, 2, (40, 6, 3)), (3, 3, (24, 12, 5))):
    G = M.Context(dim=dim, degree=p, reps=reps, hi=tuple(r / max(reps) for r in reps))
    G.set_tuning("precond", 1)
    t = (0.0, -2e3, 0.0)[:dim]
    out = []
    for k in range(3):
        G.set_interface_traction(tuple((k + 1) / 10 * x for x in t))
        t1 = time.perf_counter(); rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0); dt = time.perf_counter() - t1
        out.append((rc, info.newton_iterations, info.lin_its_total, round(1e3 * dt, 1)))
    print("MI_MG_COARSEST=%s" % os.environ.get("MI_MG_COARSEST", "(default 4)"), dim, p, reps, G.n, out, flush=True)
    G.close()
