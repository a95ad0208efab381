"""The numbers profiles/r06/README.md quotes, read back from the files of a profile set:  python tools/r6_profile_digest.py [dir]"""
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r06")


def load(name):
    return json.load(open(os.path.join(R, name)))


b = load("bench_plain_same_box_n59.json")
c, r = b["config"], b["roofline"]
print("headline: %.2f ms per step, %.2f M DoF-updates/s, %.2f CG iterations per step" % (b["ms_per_step"], b["value"] / 1e6, c["cg_iterations_per_step"]))
print("  " + ", ".join("%s %.3f" % (k, c[k]) for k in c if k.startswith("ms_")))
for k in c:
    if k.startswith("with_"):
        print("  %-52s %7.2f ms %6.2f M  %.2f its" % (k, c[k]["ms_per_step"], c[k]["value"] / 1e6, c[k]["cg_iterations_per_step"]))
print("  roofline: %.4f ms, frac %.3f, traffic %.4f GB = %.3fx, fp64 %.3f, share of step %.3f" % (
    r["avg_launch_ms"], r["frac"], (r.get("traffic") or 0) / 1e9, r.get("traffic_ratio_to_algorithmic") or 0, r["fp64"]["frac"], r["share_of_step"]))
g = r["cg_product"]
print("  cg product: %.4f ms, frac %.3f, traffic %.4f GB = %.3fx; stream-read calibration %.0f GB/s" % (
    g["avg_launch_ms"], g["frac"], (g.get("traffic") or 0) / 1e9, g.get("traffic_ratio_to_algorithmic") or 0, r["calibration_stream_read"]["GB_per_s"]))
cb = b["cpu_baseline"]
print("  cpu: (A) %.1f s = %.0f, (B) %.1f s = %.0f DoF-updates/s, GPU same mesh %.1f ms, wall %.0f s" % (
    cb["A_cg_ssor"]["t_step_s"], cb["value"], cb["B_cg_jacobi"]["t_step_s"], cb["value_cg_jacobi"], cb["gpu_same_config"]["ms_per_step"], cb["wall_s"]))
c4 = cb.get("config4_one_newton_iteration")
if c4:
    print("  cpu configuration 4: assembly %.1f s, (A) %.1f s, (B) %.1f s" % (c4["t_assembly_s"], c4["A_cg_ssor"]["t_newton_iteration_s"], c4["B_cg_jacobi"]["t_newton_iteration_s"]))
m = load("bench_plain_same_box_matrix_free_fine_level.json")
cm, rm = m["config"], m["roofline"]
print("matrix-free fine level as the process's path: %.2f ms, %.2f M; %s" % (m["ms_per_step"], m["value"] / 1e6, ", ".join("%s %.3f" % (k, cm[k]) for k in cm if k.startswith("ms_"))))
print("  mf_spmv27 %.4f ms (share %.3f), CG product %.4f ms" % (rm["avg_launch_ms"], rm["share_of_step"], rm["cg_product"]["avg_launch_ms"]))
for f in ("bench_n34_config3", "bench_n34_config3_matrix_free_fine_level", "bench_n120_42M_dofs", "bench_n120_42M_dofs_matrix_free_fine_level"):
    d = load(f + ".json")
    print("%s: %.2f ms, %.2f M, %.2f its" % (f, d["ms_per_step"], d["value"] / 1e6, d["config"]["cg_iterations_per_step"]))
for t in ("", "_matrix_free_fine_level"):
    print("emulated slabs%s:" % t, " / ".join("%.1f" % load("emulated_slabs/slabs%d%s.json" % (N, t))["ms_per_step"] for N in (1, 2, 4, 8)),
          "; weak 8: %.1f ms per slab" % (load("emulated_slabs/weak_slabs8%s.json" % t)["ms_per_step"] / 8))
for t in ("headline", "matrix_free_fine_level"):
    d = load("bench_under_rocprof_%s_n59.json" % t)
    print("under rocprof, %s: %.2f ms per step, roofline kernel %.4f ms, cg product %.4f ms" % (t, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["cg_product"]["avg_launch_ms"]))
    p = load("pmc_bench_%s_n59.json" % t)
    for k, v in p.items():
        if isinstance(v, dict) and v.get("traffic_GB_per_launch", 0) > 0.4:
            print("    %-64s %4d launches %7.4f GB" % (k[:64], v["launches"], v["traffic_GB_per_launch"]))
for f in ("rank_share_n59.txt", "smoother_quadrature_check_n59.txt", "matrix_free_fine_level_check_n59.txt", "asm_box_geometry_ab_n59.txt"):
    print("--", f)
    print(open(os.path.join(R, f)).read().rstrip()[-1400:])
for f in ("long_run_60_steps_headline.txt", "long_run_60_steps_matrix_free_fine_level.txt"):
    print("--", f, open(os.path.join(R, f)).read().rstrip().splitlines()[-2:])
