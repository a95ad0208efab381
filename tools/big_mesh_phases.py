"""Convergence of the preconditioned CG on a big mesh, with capped iterations:
python tools/big_mesh_phases.py <cells | nx,ny,nz> [key=value ...]   (environment switches of the library apply; SLABS=N)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import _pkg
M = _pkg()
reps = tuple(int(v) for v in sys.argv[1].split(",")) if "," in sys.argv[1] else (int(sys.argv[1]),) * 3
slabs = int(os.environ.get("SLABS", "1"))
G = M.Context(dim=3, degree=2, reps=reps, hi=tuple(r / max(reps) for r in reps), slabs=slabs)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    G.set_tuning(k, int(v))
G.set_interface_traction((0.0, -2e2, 0.0))
G.newton_begin_step(); G.update_acceleration(); G.assemble()
for cap in (5, 10, 20, 40):
    G.set(M.V_NEWTON, np.zeros(G.n))
    t0 = time.perf_counter(); rc, its, res = G.cg_solve(rel_tol=1e-6, max_it=cap)
    print("dofs %d cap %2d: rc %d its %d res %.3e  (%.2f s)" % (G.n, cap, rc, its, res, time.perf_counter() - t0), flush=True)
    if rc == 0:
        break
