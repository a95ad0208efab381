"""Round 6: a tangent assembly at 59^3 cells as one fused kernel ("asm_split" 0, round 5) and as point pass + tangent from
the records ("asm_split" 1), same process; the assembled values compared.  python tools/r6_asm_split.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg

M = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
G = M.Context(dim=3, degree=2, reps=(n, n, n))
h = 1.0 / n
u = 0.002 * h * np.random.default_rng(1234).standard_normal(G.n)
u[G.constrained] = 0
G.set(M.V_U, u)
G.set_interface_traction((0.0, -2e3, 0.0))
G.newton_begin_step()
G.update_acceleration()
x = np.cos(0.37 * np.arange(G.n) + 0.11)
ys = {}
for rnd in range(2):
    for split in (0, 1, 2):
        G.set_tuning("asm_split", split)
        G.assemble()
        ys[split] = G.spmv(x)
        print("asm_split", split, "ms per tangent assembly", G.bench_assemble(10), flush=True)
print("K x rel diff", np.abs(ys[1] - ys[0]).max() / np.abs(ys[0]).max(), np.abs(ys[2] - ys[0]).max() / np.abs(ys[0]).max())
