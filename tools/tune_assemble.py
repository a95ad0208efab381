#!/usr/bin/env python3
"""A/B timing of the element-kernel variants in ONE process (interleaved rounds), with a checksum cross-check.

  python tools/tune_assemble.py --cells 59 --rounds 4 --reps 3 --variants 0,1,2,3,4,5
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=59)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", type=str, default="0,1,2")
    args = ap.parse_args()
    M = _pkg()
    n = args.cells
    G = M.Context(dim=3, degree=2, reps=(n, n, n))
    G.set_tuning("precond", 0)
    G.set_interface_traction((0.0, -2e3, 0.0))
    rng = np.random.default_rng(1234)
    G.set(M.V_U, 0.02 / (2 * n) * rng.standard_normal(G.n) * (~G.constrained))
    G.update_acceleration()
    x = np.random.default_rng(4321).standard_normal(G.n)
    variants = [int(v) for v in args.variants.split(",")]
    ref = None
    for v in variants:
        G.set_tuning("asm_variant", v)
        G.assemble()
        y, r = G.spmv(x), G.get(M.V_RHS)
        if ref is None:
            ref = (y, r)
        e1 = np.abs(y - ref[0]).max() / np.abs(ref[0]).max()
        e2 = np.abs(r - ref[1]).max() / np.abs(ref[1]).max()
        print("variant %d: K.x rel diff %.2e, rhs rel diff %.2e" % (v, e1, e2), flush=True)
        assert v >= 6 or (e1 < 1e-13 and e2 < 1e-13)  # variants 6-8 are timing-only ablations
    res = {}
    for _ in range(args.rounds):
        for v in variants:
            G.set_tuning("asm_variant", v)
            res.setdefault(v, []).append(G.bench_assemble(args.reps))
    for v, ts in sorted(res.items()):
        print("variant %d: median %.2f ms  min %.2f ms per assembly (memset + 8 colours + faces + diag)"
              % (v, float(np.median(ts)), float(np.min(ts))), flush=True)


if __name__ == "__main__":
    main()
