"""Cross-check of the product forms on a big mesh: python tools/big_mesh_products.py <cells>"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import _pkg
M = _pkg()
n = int(sys.argv[1])
G = M.Context(dim=3, degree=2, reps=(n, n, n))
print("dofs", G.n, "smoother form", G.get_tuning("smoother_operator_active"), flush=True)
G.set_interface_traction((0.0, -2e2, 0.0))
G.newton_begin_step(); G.update_acceleration(); G.assemble()
x = np.random.default_rng(1).standard_normal(G.n)
ys = {}
for v in (3, 1, 4):
    G.set_tuning("spmv_variant", v)
    t0 = time.perf_counter(); ys[v] = G.spmv(x); print("variant", v, "%.2f s" % (time.perf_counter() - t0), "|y|", np.linalg.norm(ys[v]), flush=True)
for v in (1, 4):
    d = np.abs(ys[v] - ys[3]); i = int(d.argmax())
    print("variant %d vs 3: max diff %.3e (rel %.3e) at dof %d, first bad dof %s" % (v, d.max(), d.max() / np.abs(ys[3]).max(), i,
          np.flatnonzero(d > 1e-9 * np.abs(ys[3]).max())[:5]), flush=True)
