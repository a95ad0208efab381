"""Round 6: assemble_q2sf with the cells' geometry from 1/h and the volume ("asm_box_geometry" 1) against the trilinear map evaluated
at every point (0), same process, 59^3 boxes: ms per tangent assembly, per residual-only pass, the assembled values compared.
  python tools/r6_asm_box.py [n]"""
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
ys, rs = {}, {}
for rnd in range(3):
    for box in (0, 1):
        G.set_tuning("asm_box_geometry", box)
        G.assemble()
        ys[box], rs[box] = G.spmv(x), G.get(M.V_RHS)
        print("asm_box_geometry", box, "ms per tangent assembly", G.bench_assemble(10), flush=True)
print("K x rel diff", np.abs(ys[1] - ys[0]).max() / np.abs(ys[0]).max(), "rhs rel diff", np.abs(rs[1] - rs[0]).max() / np.abs(rs[0]).max())
for fine in (1,):
    G.set_tuning("fine_level", 1)
    for box in (0, 1, 0, 1):
        G.set_tuning("asm_box_geometry", box)
        G.assemble()
        print("matrix-free fine level, asm_box_geometry", box, "ms per tangent pass + diagonal blocks", G.bench_assemble(10), flush=True)
