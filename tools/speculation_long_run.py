"""Speculative enqueue of the multigrid-PCG (cg_speculate) over a long run with a load jump: the iteration tables and the final
state must be those of the polled loop bit by bit; prints the host synchronisations saved.
  python tools/speculation_long_run.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
M = importlib.import_module("dealii-adapter_amd")
res = []
for spec in (1, 0):
    G = M.Context(dim=3, degree=2, reps=(24, 24, 24))
    G.set_tuning("precond", 1); G.set_tuning("cg_warm_start", 2); G.set_tuning("cg_speculate", spec)
    G.reset_timings()
    its = []
    for s in range(40):
        G.set_interface_traction((0.0, -2e3 * min(1.0, (s + 1) / 10.0) * (1.0 if s < 25 else 0.3), 0.0))
        rc, info = G.newmark_step(tol_lin=1e-6)
        assert rc == 0
        its.append([int(info.lin_its[i]) for i in range(info.newton_iterations)])
    res.append((its, G.get(M.V_U), G.get_tuning("count_cg_host_sync"), G.get_tuning("count_cg_iterations"), G.get_tuning("count_cg_solves")))
    G.close()
print("same iteration tables:", res[0][0] == res[1][0], " bitwise same state:", np.array_equal(res[0][1], res[1][1]))
print("polls speculative %d, polled %d over %d iterations in %d solves" % (res[0][2], res[1][2], res[0][3], res[0][4]))
print(res[0][0][:4], res[0][0][24:28])
