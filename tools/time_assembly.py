import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from bench import _pkg
M = _pkg()
n = 59
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_tuning("precond", 0)
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1234)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
G.assemble()
ts = [G.bench_assemble(3) for _ in range(3)]
print(os.environ.get("MI_LAYOUT_TEST"), "assembly ms", ts)
