#!/usr/bin/env python3
"""Round 5 A/B of the matrix-free product (mf_spmv + mf_gather) at the headline size; the kernel shape is chosen by the
environment (MI_MF_TWO = 0: one cell per wavefront, 1 (default): two cells with the second cell's loads requested ahead,
3: the same at three waves per SIMD), so one process per shape; prints the time per product and a checksum of the result.
  MI_MF_TWO=0 python tools/r5_mf_ab.py [n = 59] [rounds = 5]
NEEDS profiles/r05/mf_spmv_two_cells_per_wave.patch applied to csrc/mi_kernels.hip (the two-cell kernel was measured slower
and is not in the tree): without it no MI_MF_TWO switch exists and every value runs the same one-cell kernel -- the tool
refuses to run then rather than print a label for a shape it did not measure."""
import os, sys, importlib, hashlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
M = importlib.import_module("dealii-adapter_amd")

_src = open(os.path.join(os.path.dirname(__file__), "..", "dealii-adapter_amd", "csrc", "mi_kernels.hip")).read()
if "MI_MF_TWO" not in _src:
    sys.exit("r5_mf_ab.py: the tree has no MI_MF_TWO switch -- apply profiles/r05/mf_spmv_two_cells_per_wave.patch and "
             "rebuild (make EXPERIMENTS=1) first")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
G = M.Context(dim=3, degree=2, reps=(n, n, n))
G.set_interface_traction((0.0, -2e3, 0.0))
rng = np.random.default_rng(1)
G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
G.update_acceleration()
G.assemble()
x = rng.standard_normal(G.n)
G.set_tuning("spmv_variant", 3)
y3 = G.spmv(x)
G.set_tuning("spmv_variant", 4)
y = G.spmv(x)
ts = [G.bench_spmv(20) for _ in range(rounds)]
print("MI_MF_TWO=%s: product + gather median %.4f ms  min %.4f ms; sha1 of y %s; against the assembled product %.2e" % (
    os.environ.get("MI_MF_TWO", "(default)"), np.median(ts), np.min(ts), hashlib.sha1(y.tobytes()).hexdigest()[:12],
    np.abs(y - y3).max() / np.abs(y3).max()), flush=True)
