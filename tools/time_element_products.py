"""Time one fine-level product at the headline size in its three forms (assembled sliced-ELL, element tangents,
matrix-free from quadrature-point records) in ONE process, and check the three against each other.
  python tools/time_element_products.py [n = 59] [forms = 2,1]"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
M = importlib.import_module("dealii-adapter_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
res = {}
forms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 1]
for form in forms:
    G = M.Context(dim=3, degree=2, reps=(n, n, n))
    G.set_tuning("smoother_operator", form)
    G.set_interface_traction((0.0, -2e3, 0.0))
    rng = np.random.default_rng(1)
    G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))  # 2 % of the node spacing: no cell folds
    G.update_acceleration()
    G.assemble()
    x = rng.standard_normal(G.n)
    G.set_tuning("spmv_variant", 3)
    y0 = G.spmv(x)
    t0 = G.bench_spmv(20)
    G.set_tuning("spmv_variant", 4)
    y1 = G.spmv(x)
    t1 = G.bench_spmv(20)
    print("form %d: assembled %.3f ms, element form %.3f ms, rel diff %.2e, active %d" % (
        form, t0, t1, np.abs(y1 - y0).max() / np.abs(y0).max(), G.get_tuning("smoother_operator_active")), flush=True)
    del G
