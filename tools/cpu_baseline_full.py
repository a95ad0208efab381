#!/usr/bin/env python3
"""Whole first Newmark step of BASELINE configuration 3 (34^3 Q2 cells, 985,527 DoFs) with the CPU restatement of the
reference algorithm (oracle/), pinned to one socket: (A) CG + SSOR(0.65) as the reference configures it
(nonlinear_elasticity.cc:1167-1191), (B) CG + Jacobi.  Minutes of CPU time; the result is committed as
profiles/r02/cpu_baseline_config3_full.json and supplies the per-step unit counts (assemblies, CG iterations) that
bench.py's bounded live sample is scaled with.

  python tools/cpu_baseline_full.py [cells] > gpurun_out/cpu_baseline_config3_full.json
  python tools/cpu_baseline_full.py 59 newton > gpurun_out/cpu_baseline_config4_one_newton_iteration.json
      (configuration 4: ONE Newton iteration = one assembly + one linear solve, as BASELINE.md section 2 prescribes when
       whole steps at 5 M DoFs are out of reach on the CPU)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 34
mode = 2 if len(sys.argv) > 2 and sys.argv[2] == "newton" else 1
sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "%d,0,0,%d" % (cells, mode)]).returncode)
