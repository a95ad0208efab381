"""Round 6: mid-size random 3D Q2 meshes (where the smoother multiplies matrix-free and slabs have layers to split) through the
round's new paths against the round-5 path on the same mesh: (A) one slab, assembled fine level, 64-point smoother; (B) random
slab count, assembled fine level, 27-point smoother; (C) the same slabs, fine level matrix-free, 27-point smoother.  Two Newmark
steps each; displacements and iteration tables compared.  python tools/r6_fuzz_midsize.py [first seed = 0] [minutes = 8] [other]
("other": 2D Q1-Q3 meshes of 40-120 cells and 3D Q1 meshes of 24-48 cells per direction, one slab against 2-4 slabs)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg  # noqa: E402
import oracle_lib as O  # noqa: E402  (face role constants only)

M = load_pkg()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = 60.0 * float(sys.argv[2]) if len(sys.argv) > 2 else 480.0
other = len(sys.argv) > 3 and sys.argv[3] == "other"  # 2D Q1-Q3 and 3D Q1 meshes instead: one slab against a random slab count
t0 = time.time()
seed, done, worst, bad = first, 0, 0.0, []
while other and time.time() - t0 < budget:
    rng = np.random.default_rng(9000 + seed)
    dim = 2 if rng.random() < 0.6 else 3
    p = int(rng.integers(1, 4)) if dim == 2 else 1
    reps = tuple(int(rng.integers(40, 121)) for _ in range(2)) if dim == 2 else tuple(int(rng.integers(24, 49)) for _ in range(3))
    h = rng.uniform(0.02, 0.05, dim)
    hi = tuple(float(h[d] * reps[d]) for d in range(dim))
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, int(rng.choice([0, O.FACE_CLAMPED, O.FACE_INTERFACE])), O.FACE_INTERFACE]
    if dim == 3:
        roles += [int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE])), int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE]))]
    else:
        roles += [0, 0]
    slabs = int(rng.integers(2, 5))
    kw = dict(mu=float(10 ** rng.uniform(5, 6.5)), nu=float(rng.uniform(0.2, 0.45)), rho=float(rng.uniform(500, 2000)), delta_t=0.005)
    trac = (0.0, -float(10 ** rng.uniform(2, 3.3)), float(rng.uniform(-200, 200)))[:dim]
    out = {}
    for tag, s in (("A", 1), ("B", slabs)):
        G = M.Context(dim=dim, degree=p, reps=reps, hi=hi, face_role=roles, slabs=s, **kw)
        G.set_tuning("precond", 1)
        G.set_tuning("cg_warm_start", 2)
        rows = []
        for k in range(2):
            G.set_interface_traction(tuple(t * (k + 1) / 2 for t in trac))
            rc, info = G.newmark_step(tol_lin=1e-8)
            rows.append((rc, info.newton_iterations, info.lin_its_total))
        out[tag] = (rows, G.get(M.V_U))
        G.close()
    d = np.abs(out["B"][1] - out["A"][1]).max() / np.abs(out["A"][1]).max()
    worst = max(worst, d)
    if not (d < 1e-6 and all(a[0] == 0 and b[0] == 0 and a[1] == b[1] and abs(a[2] - b[2]) <= 2 for a, b in zip(out["A"][0], out["B"][0]))):
        bad.append((seed, dim, p, reps, slabs, d, out["A"][0], out["B"][0]))
        print("MISMATCH", bad[-1], flush=True)
    done += 1
    seed += 1
while not other and time.time() - t0 < budget:
    rng = np.random.default_rng(7000 + seed)
    reps = tuple(int(rng.integers(23, 35)) for _ in range(3))
    h = rng.uniform(0.02, 0.05, 3)
    hi = tuple(float(h[d] * reps[d]) for d in range(3))
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, int(rng.choice([0, O.FACE_CLAMPED, O.FACE_INTERFACE])), O.FACE_INTERFACE,
             int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE])), int(rng.choice([O.FACE_ZCLAMP, O.FACE_INTERFACE]))]
    perturb = None
    if rng.random() < 0.4:
        perturb = 0.08 * h.min() * rng.uniform(-1, 1, (int(np.prod([r + 1 for r in reps])), 3))
    slabs = int(rng.integers(1, 5))
    kw = dict(mu=float(10 ** rng.uniform(5, 6.5)), nu=float(rng.uniform(0.2, 0.45)), rho=float(rng.uniform(500, 2000)), delta_t=0.005)
    trac = (0.0, -float(10 ** rng.uniform(2, 3.3)), float(rng.uniform(-200, 200)))
    out = {}
    for tag, s, fine, quad in (("A", 1, 0, 4), ("B", slabs, 0, 3), ("C", slabs, 1, 3)):
        G = M.Context(dim=3, degree=2, reps=reps, hi=hi, face_role=roles, perturb=perturb, slabs=s, **kw)
        G.set_tuning("precond", 1)
        G.set_tuning("cg_warm_start", 2)
        G.set_tuning("smoother_quadrature", quad)
        if fine:
            G.set_tuning("fine_level", 1)
            G.set_tuning("mf_diag_lag", int(rng.integers(0, 2)))
        rows = []
        for k in range(2):
            G.set_interface_traction(tuple(t * (k + 1) / 2 for t in trac))
            rc, info = G.newmark_step(tol_lin=1e-8)
            rows.append((rc, info.newton_iterations, info.lin_its_total))
        out[tag] = (rows, G.get(M.V_U), G.get_tuning("smoother_quadrature_active"), G.get_tuning("smoother_operator_active"))
        G.close()
    scale = np.abs(out["A"][1]).max()
    for tag in ("B", "C"):
        d = np.abs(out[tag][1] - out["A"][1]).max() / scale
        worst = max(worst, d)
        ok = d < 1e-6 and all(a[0] == 0 and b[0] == 0 and a[1] == b[1] and abs(a[2] - b[2]) <= 2 for a, b in zip(out["A"][0], out[tag][0]))
        if not ok:
            bad.append((seed, tag, reps, slabs, d, out["A"][0], out[tag][0]))
            print("MISMATCH", bad[-1], flush=True)
    done += 1
    if done % 10 == 0:
        print("seed %d: reps %s slabs %d distorted %s smoother %s/%s, worst displacement difference so far %.2e" % (
            seed, reps, slabs, perturb is not None, out["C"][2], out["C"][3], worst), flush=True)
    seed += 1
print("%d meshes (seeds %d..%d) in %.0f s: worst relative displacement difference %.2e, mismatches %d" % (done, first, seed - 1, time.time() - t0, worst, len(bad)))
