"""Kernel timeline of the shipped FSI3 case (2D, degree 3, 1,100 dofs) with `Solver type = Direct`: run under
rocprofv3 --kernel-trace --stats to see where the milliseconds of a Newmark step go.
  python tools/direct_case_trace.py [steps = 20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg  # noqa: E402
M = _pkg()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = M.Context(dim=2, degree=3, reps=(18, 3), lo=(0.24899, 0.19), hi=(0.6, 0.21), face_role=[1, 7, 7, 7, 8, 8])
G.set_tuning("solver_type", 1)
for k in range(3):
    G.set_interface_traction((0.0, -40.0 * min(1.0, (k + 1) / 10)))
    G.newmark_step(tol_lin=1e-6, max_it_mult=10.0)
t0 = time.perf_counter()
for k in range(3, 3 + steps):
    G.set_interface_traction((0.0, -40.0 * min(1.0, (k + 1) / 10)))
    rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=10.0)
    assert rc == 0
print("FSI3 2D p=3 direct: %.3f ms per step, %d Newton iterations in the last step" % (
    1e3 * (time.perf_counter() - t0) / steps, info.newton_iterations))
