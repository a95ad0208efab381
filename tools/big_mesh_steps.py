"""Newmark steps on a big 3D Q2 block in one context: python tools/big_mesh_steps.py <cells> [tuning_key=value ...]"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import _pkg
M = _pkg()
n = int(sys.argv[1])
t0 = time.perf_counter()
G = M.Context(dim=3, degree=2, reps=(n, n, n))
for kv in sys.argv[2:]:  # tuning keys, e.g. smoother_operator=0 asm_variant=9
    k, v = kv.split("=")
    G.set_tuning(k, int(v))
print("setup %.1f s, dofs %d" % (time.perf_counter() - t0, G.n), flush=True)
for k in range(3):
    G.set_interface_traction((0.0, -2e3 * (k + 1) / 10, 0.0))
    t0 = time.perf_counter(); rc, info = G.newmark_step(tol_lin=1e-6, max_it_mult=1.0); dt = time.perf_counter() - t0
    print("step", k, "rc", rc, "newton", info.newton_iterations, "cg", info.lin_its_total, "%.1f ms" % (1e3 * dt), "%.2f M DoF/s" % (G.n / dt / 1e6), flush=True)
