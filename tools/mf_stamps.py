"""Where a wavefront of mf_spmv spends its life (MI_MF_STAMPS diagnostic), lattice ids on / off.
  MI_MF_STAMPS=1 python tools/mf_stamps.py [n = 59]"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["MI_MF_STAMPS"] = "1"
M = importlib.import_module("dealii-adapter_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
G.assemble()
G.set_tuning("spmv_variant", 4)
for lat in (1, 0):
    G.set_tuning("cell_lattice", lat)
    print("cell_lattice", lat, "product %.4f ms" % G.bench_spmv(20), flush=True)
