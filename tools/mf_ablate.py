"""Timing-only ablations of mf_spmv (MI_MF_DBG: 2 no result stores, 4 records of cell 0 for every cell, 8 x of cell 0's nodes,
6 = 2+4, 14 = all three): which stream the product waits for.  One process per setting (the switch is read once).
  python tools/mf_ablate.py [n = 59]"""
import os, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "59"
code = r'''
import os, sys, importlib
import numpy as np
sys.path.insert(0, %r)
M = importlib.import_module("dealii-adapter_amd")
n = int(%r)
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
G.assemble()
G.set_tuning("spmv_variant", 4)
t = [G.bench_spmv(20) for _ in range(4)]
print("MI_MF_DBG=%%s: product (launch + gather) %%.4f ms (min %%.4f)" %% (os.environ.get("MI_MF_DBG", "0"), np.median(t), np.min(t)), flush=True)
''' % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), n)
for dbg in ("0", "2", "4", "8", "6", "14"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MI_MF_DBG=dbg))
