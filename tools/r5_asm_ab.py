#!/usr/bin/env python3
"""Round 5 A/B of the 3D Q2 element kernel (assemble_q2sf) in ONE process at the headline size: the variants named on
the command line against the default, bit-for-bit comparison of K.x and the residual, interleaved timing rounds, and the
phase stamps of every variant.
  python tools/r5_asm_ab.py [n = 59] [rounds = 4] [variants = 3,4,5]"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
M = importlib.import_module("dealii-adapter_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variants = [0] + [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "3,4,5").split(",")]
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
x = rng.standard_normal(G.n)
ref, res = None, {}
for v in variants:
    G.set_tuning("asm_variant", v)
    G.assemble()
    y, r = G.spmv(x), G.get(M.V_RHS)
    if ref is None:
        ref = (y, r)
    print("asm_variant %d: K.x bitwise %s (rel %.1e), rhs bitwise %s (rel %.1e)" % (
        v, np.array_equal(y, ref[0]), np.abs(y - ref[0]).max() / np.abs(ref[0]).max(), np.array_equal(r, ref[1]),
        np.abs(r - ref[1]).max() / np.abs(ref[1]).max()), flush=True)
for _ in range(rounds):
    for v in variants:
        G.set_tuning("asm_variant", v)
        res.setdefault(v, []).append(G.bench_assemble(3))
for v, t in res.items():
    print("asm_variant %d: median %.3f ms  min %.3f ms per tangent assembly" % (v, np.median(t), np.min(t)), flush=True)
os.environ["MI_ASM_STAMPS"] = "1"
for v in variants:
    G.set_tuning("asm_variant", v)
    print("phase stamps, asm_variant %d:" % v, flush=True)
    sys.stderr.flush()
    G.bench_assemble(1)
    sys.stderr.flush()
