#!/usr/bin/env python3
"""A/B timing of the block-CSR SpMV variants in ONE process, interleaved rounds (median and min reported).

  python tools/tune_spmv.py --cells 59 --rounds 5 --reps 20
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _pkg, spmv_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=59)
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--grids", type=str, default="1280,2048,4096")
    ap.add_argument("--variants", type=str, default="1,3,13", help="3 = sliced-ELL (reported as 31..34 by unroll)")
    ap.add_argument("--unrolls", type=str, default="1,2,4", help="sliced-ELL unrolls; -1,-2,-3 are timing-only ablations")
    args = ap.parse_args()
    global UNROLLS
    UNROLLS = [int(u) for u in args.unrolls.split(",")]
    M = _pkg()
    n = args.cells
    G = M.Context(dim=3, degree=args.degree, reps=(n, n, n))
    G.set_interface_traction((0.0, -2e3, 0.0))
    rng = np.random.default_rng(1234)
    G.set(M.V_U, 0.02 / (n * args.degree) * rng.standard_normal(G.n) * (~G.constrained))
    G.update_acceleration()
    G.assemble()
    x = np.random.default_rng(4321).standard_normal(G.n)
    variants = [int(v) for v in args.variants.split(",")]
    grids = [int(g) for g in args.grids.split(",")]
    # cross-check the variants against variant 0
    G.set_tuning("spmv_variant", 1)
    y0 = G.spmv(x)
    for v in variants:
        if v > 3:  # timing-only ablations (11, 12) and stream kernels (13, 14) produce no valid y
            continue
        G.set_tuning("spmv_variant", v)
        y = G.spmv(x)
        err = np.abs(y - y0).max() / np.abs(y0).max()
        print("variant %d vs 1: max rel diff %.2e" % (v, err), flush=True)
        assert err < 1e-13
    nbytes = spmv_bytes(G.nnodes, G.nnz // 9, 3)
    res = {}
    for r in range(args.rounds):
        for v in variants:
            for g in grids:
                for u in (UNROLLS if v == 3 else [0]):
                    G.set_tuning("spmv_variant", v)
                    if g:  # 0: the library's own launch geometry
                        G.set_tuning("spmv_grid", g)
                    if u:
                        G.set_tuning("sell_unroll", u)
                    res.setdefault((v * 100 + u if v == 3 else v, g), []).append(G.bench_spmv(args.reps))
    out = []
    for (v, g), ts in sorted(res.items()):
        med, mn = float(np.median(ts)), float(np.min(ts))
        out.append({"variant": v, "grid": g, "median_ms": med, "min_ms": mn, "GBs_median": nbytes / med / 1e6,
                    "frac_of_8TBs": nbytes / med / 1e6 / 8000.0})
        print("variant %d grid %5d: median %.3f ms  min %.3f ms  -> %.0f GB/s (%.1f %% of 8 TB/s)" %
              (v, g, med, mn, nbytes / med / 1e6, 100 * nbytes / med / 1e6 / 8000.0), flush=True)
    print(json.dumps({"n_dofs": G.n, "bytes_per_launch": nbytes, "results": out}))


if __name__ == "__main__":
    main()
