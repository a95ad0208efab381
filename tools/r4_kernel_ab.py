#!/usr/bin/env python3
"""Round 4 A/B of the two hot kernels in ONE process at the headline size: node ids by arithmetic (mi::CellLattice) against
the connectivity load in mf_spmv and assemble_q2sf, and the assemble_q2sf experiments (asm_variant 3: rotating prologue
wave, 4: L2 atomics for later touches, 5: both).  Results must agree bit by bit.
  python tools/r4_kernel_ab.py [n = 59] [rounds = 4]"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
M = importlib.import_module("dealii-adapter_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
print("cell_lattice available:", G.get_tuning("cell_lattice"), flush=True)
x = rng.standard_normal(G.n)

# ---- matrix-free product
G.assemble()
G.set_tuning("spmv_variant", 4)
ys, ts = {}, {0: [], 1: []}
for lat in (1, 0):
    G.set_tuning("cell_lattice", lat)
    ys[lat] = G.spmv(x)
for _ in range(rounds):
    for lat in (1, 0):
        G.set_tuning("cell_lattice", lat)
        ts[lat].append(G.bench_spmv(20))
print("mf product (launch + gather): lattice %.4f ms (min %.4f)  conn %.4f ms (min %.4f)  bitwise equal: %s" % (
    np.median(ts[1]), np.min(ts[1]), np.median(ts[0]), np.min(ts[0]), np.array_equal(ys[0], ys[1])), flush=True)
G.set_tuning("spmv_variant", 3)
y3 = G.spmv(x)
print("  against the assembled product: rel diff %.2e" % (np.abs(ys[1] - y3).max() / np.abs(y3).max()), flush=True)

# ---- element kernel
cases = [(0, 1), (0, 0), (3, 1), (4, 1), (5, 1)]
ref, res = None, {}
for v, lat in cases:
    G.set_tuning("cell_lattice", lat)
    G.set_tuning("asm_variant", v)
    G.assemble()
    y, r = G.spmv(x), G.get(M.V_RHS)
    if ref is None:
        ref = (y, r)
    print("asm_variant %d lattice %d: K.x bitwise %s (rel %.1e), rhs bitwise %s" % (
        v, lat, np.array_equal(y, ref[0]), np.abs(y - ref[0]).max() / np.abs(ref[0]).max(), np.array_equal(r, ref[1])), flush=True)
for _ in range(rounds):
    for v, lat in cases:
        G.set_tuning("cell_lattice", lat)
        G.set_tuning("asm_variant", v)
        res.setdefault((v, lat), []).append(G.bench_assemble(3))
for (v, lat), t in res.items():
    print("asm_variant %d lattice %d: median %.3f ms  min %.3f ms per tangent assembly" % (v, lat, np.median(t), np.min(t)), flush=True)
