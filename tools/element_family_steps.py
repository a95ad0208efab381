"""Two Newmark steps with the default solver over element families and mesh extremes: python tools/element_family_steps.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import _pkg
M = _pkg()
for dim, p, reps in ((3, 3, (24, 24, 24)), (3, 4, (14, 14, 14)), (3, 1, (150, 150, 150)), (2, 2, (1500, 1500)), (3, 2, (2, 2, 2)), (3, 2, (1, 1, 1)), (3, 3, (40, 6, 2))):
    t0 = time.perf_counter()
    G = M.Context(dim=dim, degree=p, reps=reps)
    t = (0.0, -2e3, 0.0)[:dim]
    out = []
    for k in range(2):
        G.set_interface_traction(tuple((k + 1) / 10 * x for x in t))
        t1 = time.perf_counter(); rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0); dt = time.perf_counter() - t1
        out.append((rc, info.newton_iterations, info.lin_its_total, round(1e3 * dt, 1)))
    print(dim, p, reps, G.n, out, "setup+2 steps %.1f s" % (time.perf_counter() - t0), flush=True)
    G.close()
