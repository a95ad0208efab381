#!/usr/bin/env python3
"""bench.py -- DoF-updates/s of one Newmark step (Newton x (assembly + CG) + Newmark updates) of the
neo-Hookean solver on a synthetic 3D Q2 block (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

works as a bare command for every N: with N > 1 and no WORLD_SIZE in the environment it starts the N ranks itself
as child processes (`python -m torch.distributed.run ...`, before anything touches the GPU) and relays rank 0's
JSON line; under an external `torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

A step = one full Newmark step on the resident mesh under the ramped constant interface traction
(0,-2e3,0) Pa (ramp over the first 10 steps).  Rank 0 prints ONE JSON line.

N > 1: the headline workload is BASELINE configuration 4 as written -- the ONE 59^3-cell block (5,055,477 DoFs)
domain-decomposed over the N GPUs ("scaling": "strong"); the weak-scaling number (one 59^3 block per GPU) is
measured afterwards and reported as a second field, `weak_scaling`.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP64_VECTOR_PEAK_TF = 78.6  # MI355X FP64 vector (non-MFMA) peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
MF_FLOPS_PER_LANE = 712.0  # FP64 flop per lane of a cell's wavefront in mf_spmv: (2 x 61.61 M FMA + 20.54 M MUL + 2.46 M ADD
# wave instructions) x 64 lanes / (205,379 cells x 64 lanes), profiles/r05/pmc_counters_mf_spmv_n59.json
MF27_FLOPS_PER_LANE = 642.0  # mf_spmv27, per lane of a wavefront of TWO cells: 2 x 267 FMA + 96 MUL + 12 ADD FP64 wave instructions
CPU_FULL_RUN = os.path.join(ROOT, "profiles", "r02", "cpu_baseline_config3_full.json")  # unit counts of a whole CPU step (deterministic)


def _pkg():
    name = "dealii_adapter_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "dealii-adapter_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def spmv_bytes(nnodes, nnzb, dim, generated_columns=None):
    """algorithmic HBM bytes of one SpMV launch of the format actually used: every stored value once, the block
    column indices once -- or, with generated columns (the default, `sell_icol`), 8 bytes of column-box description
    per row instead --, 4 bytes of row bookkeeping per row, x read once, y written once (the fused dot's second read
    of p is not credited)"""
    if generated_columns is None:
        generated_columns = True  # the library's default ("sell_icol" 1)
    n = nnodes * dim
    index_bytes = 8 * nnodes if generated_columns else 4 * nnzb
    return 8 * nnzb * dim * dim + index_bytes + 4 * (nnodes + 1) + 8 * n + 8 * n


def spmv_bytes_scalar_csr(nnodes, nnzb, dim):
    """SURVEY.md section 8(d) figure for a scalar CSR of the same matrix: 12*nnz + 4*(N+1) + 16*N"""
    n = nnodes * dim
    return 12 * nnzb * dim * dim + 4 * (n + 1) + 16 * n


# ---------------------------------------------------------------------------------------------- CPU baseline
def socket0_cores():
    """one logical CPU per physical core of the first socket, restricted to the CPUs this process may use;
    returns (cpus, model name)"""
    allowed = sorted(os.sched_getaffinity(0))
    pkg_of, first_sibling = {}, {}
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            pkg_of[c] = int(open(base + "physical_package_id").read())
            sib = open(base + "thread_siblings_list").read().strip()
            first_sibling[c] = int(sib.replace("-", ",").split(",")[0])
        except (OSError, ValueError):
            pkg_of[c], first_sibling[c] = 0, c
    pkg0 = min(pkg_of.values())
    cpus = sorted({first_sibling[c] for c in allowed if pkg_of[c] == pkg0 and first_sibling[c] in pkg_of})
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cpus or allowed, model


def cpu_baseline_worker(cells, its_a, its_b, full):
    """runs in a CHILD process that never touches the GPU: pins itself to one socket (one thread per physical core)
    BEFORE the oracle's OpenMP runtime starts, then times the CPU restatement of the reference algorithm on BASELINE
    configuration 3 (cells = 34: 985,527 DoFs).  Prints one JSON object.
      full = 0: bounded sample -- one assembly, `its_a` CG+SSOR(0.65) iterations, `its_b` CG+Jacobi iterations
      full = 1: whole first Newmark step with (A) CG+SSOR(0.65) and (B) CG+Jacobi (minutes; tools/cpu_baseline_full.py)"""
    cpus, model = socket0_cores()
    os.sched_setaffinity(0, cpus)
    os.environ["OMP_PROC_BIND"] = "close"
    os.environ["OMP_PLACES"] = "cores"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.lib().orc_set_threads(len(cpus))
    out = {"cells": cells, "cores": len(cpus), "threads": len(cpus), "cpu_model": model,
           "pinned_cpus": "%d-%d" % (cpus[0], cpus[-1]) if cpus == list(range(cpus[0], cpus[-1] + 1)) else cpus}

    def fresh():
        P = O.Problem(O.make_desc(dim=3, degree=2, reps=(cells,) * 3))
        P.set_interface_traction((0.0, -2e2, 0.0))  # first ramp step
        return P

    P = fresh()
    out["n_dofs"], out["nnz"] = int(P.n), int(P.nnz)
    if full == 2:
        # ONE Newton iteration (one assembly + one linear solve from zero) of the first step: what BASELINE.md section 2
        # prescribes for configuration 4 (59^3 cells) when whole steps are out of reach on the CPU
        P.update_acceleration()
        t0 = time.perf_counter()
        P.assemble()
        out["t_assembly_s"] = time.perf_counter() - t0
        print(json.dumps(out), file=sys.stderr, flush=True)
        for key, solver in (("B_cg_jacobi", O.SOLVER_CG_JACOBI), ("A_cg_ssor", O.SOLVER_CG_SSOR)):  # the short one first
            P.vec(O.V_NEWTON)[:] = 0.0
            t0 = time.perf_counter()
            rc, its, res = P.solve_linear(solver, tol_lin=1e-6, max_it_mult=1.0)
            dt = time.perf_counter() - t0
            out[key] = {"rc": rc, "cg_iterations": its, "t_solve_s": dt, "t_newton_iteration_s": out["t_assembly_s"] + dt}
            print(json.dumps(out), file=sys.stderr, flush=True)
    elif full:
        for key, solver in (("A_cg_ssor", O.SOLVER_CG_SSOR), ("B_cg_jacobi", O.SOLVER_CG_JACOBI)):
            P = fresh()
            t0 = time.perf_counter()
            rc, info = P.newmark_step(solver, tol_lin=1e-6, max_it_mult=1.0)
            dt = time.perf_counter() - t0
            out[key] = {"rc": rc, "t_step_s": dt, "t_assemble_s": info.t_assemble, "t_solve_s": info.t_solve,
                        "newton_iterations": info.newton_iterations, "assemblies": info.assemblies,
                        "cg_iterations": info.lin_its_total, "dof_updates_per_s": P.n / dt}
            print(json.dumps(out), file=sys.stderr, flush=True)
    else:
        P.update_acceleration()
        t0 = time.perf_counter()
        P.assemble()
        out["t_assembly_s"] = time.perf_counter() - t0
        for key, solver, its in (("A_cg_ssor", O.SOLVER_CG_SSOR, its_a), ("B_cg_jacobi", O.SOLVER_CG_JACOBI, its_b)):
            P.vec(O.V_NEWTON)[:] = 0.0
            t0 = time.perf_counter()
            _, done, _ = P.solve_linear(solver, tol_lin=1e-30, max_it_mult=(its + 0.5) / P.n)
            out[key] = {"iterations_timed": done, "t_per_iteration_s": (time.perf_counter() - t0) / max(done, 1)}
    print(json.dumps(out))


def _run_cpu_worker(spec, timeout_s):
    """one pinned CPU child; returns (last complete JSON object or None, last partial object from stderr or None, note)"""
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", spec], capture_output=True, text=True,
                           timeout=max(1.0, timeout_s))
        out, err, note = p.stdout, p.stderr, "" if p.returncode == 0 else "exit code %d: %s" % (p.returncode, (p.stderr or "")[-200:])
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        note = "stopped by the wall-clock guard after %.0f s" % timeout_s
    full = [l for l in out.splitlines() if l.startswith("{")]
    part = [l for l in err.splitlines() if l.startswith("{")]
    return (json.loads(full[-1]) if full else None), (json.loads(part[-1]) if part else None), note


def cpu_baseline(cells, its_a, its_b, budget_s, config4):
    """CPU leg of the bench line, measured LIVE in a child pinned to one socket:
      * BASELINE configuration 3 (cells = 34): the WHOLE first Newmark step with (A) CG+SSOR(0.65), as the reference
        configures it, and (B) CG+Jacobi -- ~200 s on a 64-core socket;
      * BASELINE configuration 4 (59^3 cells, the metric's own mesh): ONE Newton iteration (one assembly + one linear
        solve from zero with (A) and with (B)) -- ~310 s.
    `budget_s` (MI_BENCH_CPU_BUDGET_S, default 900) is a wall-clock guard over both: a leg that would not fit is
    replaced by the bounded sample of round 3 (one assembly, its_a / its_b iterations, scaled with the step's unit
    counts from the committed full run) resp. skipped, and `sample` says so.  Nothing here is read from a committed file
    unless the guard fired."""
    t_begin = time.time()
    left = lambda: budget_s - (time.time() - t_begin)
    small = cells <= 12  # seconds of CPU work
    full, part, note = _run_cpu_worker("%d,%d,%d,1" % (cells, its_a, its_b), left() if not small else 600)
    label = "3 (34^3 Q2 cells)" if cells == 34 else "%d^3 Q2 cells" % cells
    if full and full.get("A_cg_ssor", {}).get("rc") == 0 and full.get("B_cg_jacobi", {}).get("rc") == 0:
        s, a, b = full, full["A_cg_ssor"], full["B_cg_jacobi"]
        out = {"value": a["dof_updates_per_s"], "value_cg_jacobi": b["dof_updates_per_s"], "unit": "DoF-updates/s",
               "cores": s["cores"], "threads": s["threads"], "kind": "port", "cpu_model": s["cpu_model"],
               "pinned_cpus": s["pinned_cpus"], "config": label, "n_dofs": s["n_dofs"], "live": True,
               "A_cg_ssor": a, "B_cg_jacobi": b,
               "sample": "LIVE in this run: the whole first Newmark step of BASELINE configuration %s = 3D Q2 block %d^3 cells (%d DoFs, "
                         "%d nnz) on ONE pinned socket (%s, %d cores, %d threads): (A) CG+SSOR(0.65) as the reference configures it "
                         "(nonlinear_elasticity.cc:1180-1182; sweeps serial as in deal.II) %d Newton iterations / %d assemblies / %d CG "
                         "iterations in %.1f s (assembly %.1f s, solves %.1f s), (B) CG+Jacobi %d / %d / %d in %.1f s; assembly threaded "
                         "over cells with ordered scatter; restatement of the reference algorithm (oracle/), not the deal.II binary"
                         % (label, cells, s["n_dofs"], s["nnz"], s["cpu_model"], s["cores"], s["threads"], a["newton_iterations"],
                            a["assemblies"], a["cg_iterations"], a["t_step_s"], a["t_assemble_s"], a["t_solve_s"],
                            b["newton_iterations"], b["assemblies"], b["cg_iterations"], b["t_step_s"])}
    else:
        # the guard fired (or the child failed): the bounded sample of round 3, scaled with committed unit counts
        s, _, note2 = _run_cpu_worker("%d,%d,%d,0" % (cells, its_a, its_b), max(120.0, left()))
        if not s:
            return {"value": None, "unit": "DoF-updates/s", "cores": 0, "kind": "port", "live": False,
                    "sample": "CPU leg failed: full step: %s; sample: %s" % (note, note2)}
        ta, tia, tib = s["t_assembly_s"], s["A_cg_ssor"]["t_per_iteration_s"], s["B_cg_jacobi"]["t_per_iteration_s"]
        out = {"unit": "DoF-updates/s", "cores": s["cores"], "kind": "port", "threads": s["threads"], "cpu_model": s["cpu_model"],
               "pinned_cpus": s["pinned_cpus"], "config": label, "n_dofs": s["n_dofs"], "live": False, "t_assembly_s": ta,
               "t_cg_ssor_iteration_s": tia, "t_cg_jacobi_iteration_s": tib, "value": None}
        what = ("FALLBACK (the whole-step CPU leg did not fit the wall-clock guard of %.0f s: %s): bounded sample of configuration %s "
                "on ONE pinned socket (%s, %d cores): 1 assembly %.1f s, %d CG+SSOR iterations %.2f s each, %d CG+Jacobi iterations "
                "%.3f s each" % (budget_s, note, label, s["cpu_model"], s["cores"], ta, s["A_cg_ssor"]["iterations_timed"], tia,
                                 s["B_cg_jacobi"]["iterations_timed"], tib))
        if os.path.exists(CPU_FULL_RUN) and json.load(open(CPU_FULL_RUN)).get("cells") == cells:
            c = json.load(open(CPU_FULL_RUN))
            a, b = c["A_cg_ssor"], c["B_cg_jacobi"]
            out["value"] = s["n_dofs"] / (a["assemblies"] * ta + a["cg_iterations"] * tia)
            out["value_cg_jacobi"] = s["n_dofs"] / (b["assemblies"] * ta + b["cg_iterations"] * tib)
            what += ("; scaled with the unit counts of the committed full run %s ((A) %d assemblies + %d iterations, (B) %d + %d)"
                     % (os.path.relpath(CPU_FULL_RUN, ROOT), a["assemblies"], a["cg_iterations"], b["assemblies"], b["cg_iterations"]))
        out["sample"] = what + "; restatement of the reference algorithm (oracle/), not the deal.II binary"
    if config4 and not small:
        # the metric's own mesh: one Newton iteration, with what is left of the budget
        if left() < 60:
            out["config4_one_newton_iteration"] = {"live": False, "note": "skipped: %.0f s of the CPU budget left" % left()}
        else:
            f, part4, note4 = _run_cpu_worker("59,0,0,2", left())
            got = f or part4  # (the child reports after every phase on stderr: a guard that fires late keeps the phases done)
            if got:
                got["live"] = True
                got["complete"] = bool(f)
                if note4:
                    got["note"] = note4
                out["config4_one_newton_iteration"] = got
            else:
                out["config4_one_newton_iteration"] = {"live": False, "note": note4 or "no output"}
    out["wall_s"] = time.time() - t_begin
    out["budget_s"] = budget_s
    return out


# ---------------------------------------------------------------------------------------------- HBM traffic (PMC counters)
PMC_COUNTERS = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "WRITE_SIZE")


def pmc_reduce(root):
    """per kernel (name + grid): launches and HBM bytes per launch from the per-counter rocprofv3 passes under `root`.
    gfx950 corrections of the MI355X guide (HBM / rocprofv3 section; checked on the streaming calibration kernel in round 1):
    read bytes = TCC_EA0_RDREQ_sum x 128 B - TCC_EA0_RDREQ_32B_sum x 96 B, write bytes = WRITE_SIZE x 1 KiB"""
    import csv
    import glob
    import re
    from collections import defaultdict
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", row["Kernel_Name"]))
                acc["%s grid=%s" % (name, row.get("Grid_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        if not all(c in cs for c in PMC_COUNTERS):
            continue
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        rd = c["TCC_EA0_RDREQ_sum"] * 128.0 - c["TCC_EA0_RDREQ_32B_sum"] * 96.0
        out[k] = {"launches": max(len(v) for v in cs.values()), "read_bytes": rd, "write_bytes": c["WRITE_SIZE"] * 1024.0,
                  "bytes_per_launch": rd + c["WRITE_SIZE"] * 1024.0}
    return out


def pmc_traffic(cells, budget_s):
    """LIVE in the default run (round 6; `roofline.traffic` was null until round 5): one Newmark step of the headline
    configuration in three CHILD processes under `rocprofv3 --pmc <counter>` -- one counter per pass, nothing but --pmc, as the
    MI355X guide prescribes -- started while this process has not touched the GPU.  Returns (per-kernel dict or None, note)."""
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    t0 = time.time()
    root = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    me = os.path.abspath(__file__)
    for c in PMC_COUNTERS:
        left = budget_s - (time.time() - t0)
        if left < 30:
            return None, "wall-clock guard (%.0f s) reached before counter %s" % (budget_s, c)
        cmd = ["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", os.path.join(root, c), "--", "python3", me, "--pmc-child",
               "--cells", str(cells), "--cpu-cells", "0"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=left)
        except subprocess.TimeoutExpired:
            return None, "pass %s stopped by the wall-clock guard" % c
        if r.returncode != 0:
            return None, "pass %s: exit code %d: %s" % (c, r.returncode, (r.stderr or "")[-300:])
    red = pmc_reduce(root)
    shutil.rmtree(root, ignore_errors=True)
    return (red or None), ("three passes in %.0f s" % (time.time() - t0) if red else "no counter rows found")


# ---------------------------------------------------------------------------------------------- GPU side
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """start the N ranks as child processes of this (GPU-free) process and pass their exit code on.  A wall-clock limit
    (MI_BENCH_RANKS_TIMEOUT_S, default 1800 s) stands over them: ranks that hang -- an RCCL rendezvous that never completes
    on the first real multi-GPU run, say -- are ended as a process group (they were started in a session of their own) and
    the bench exits with 124 instead of hanging.  Never a re-exec: this process has not touched the GPU and only waits."""
    import signal
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    limit = float(os.environ.get("MI_BENCH_RANKS_TIMEOUT_S", "1800"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print("bench.py: the %d ranks did not finish within %.0f s (MI_BENCH_RANKS_TIMEOUT_S): ending their process group"
              % (n, limit), file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)  # exactly the group started above
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=59, help="cells per side of the Q2 block (59 -> 5,055,477 DoFs)")
    ap.add_argument("--tol-lin", type=float, default=1e-6)
    ap.add_argument("--precond", choices=["mg", "jacobi"], default="mg",
                    help="CG preconditioner: geometric multigrid V-cycle (default) or Jacobi")
    ap.add_argument("--precond-storage", choices=["f64", "f32"], default="f64",
                    help="f32: the multigrid smoother multiplies with an fp32-rounded copy of the level matrices (the CG's "
                         "own product, residuals, vectors and all arithmetic stay fp64); opt-in, not the headline setting")
    ap.add_argument("--smoother-precision", choices=["f64", "f32"], default="f64",
                    help="f32 (opt-in, not the headline setting): the multigrid smoother's matrix-free fine-level products in fp32 "
                         "arithmetic on fp32 point records; the CG, its product, residuals, vectors, assembly and every other kernel "
                         "stay fp64 (a preconditioner-only change: same stopping rule, same converged solution)")
    ap.add_argument("--smoother-operator", choices=["matrix-free", "element", "assembled"], default="matrix-free",
                    help="what the multigrid smoother multiplies with on the fine level: the unassembled symmetric element "
                         "tangents (default where available: undecomposed 3D Q2 meshes; 27 %% fewer bytes per product) or the "
                         "assembled sliced-ELL matrix; the CG's own product always uses the assembled matrix")
    ap.add_argument("--cg-operator", choices=["assembled", "element"], default="assembled",
                    help="A/B: the CG's own product on the assembled sliced-ELL matrix (default; the kernel the north star names) or "
                         "on the unassembled element tangents like the smoother's (then no sliced-ELL copy is made)")
    ap.add_argument("--smoother-quadrature", type=int, choices=[3, 4], default=3,
                    help="Gauss points per direction of the multigrid smoother's fine-level operator: 3 (the library's default since "
                         "round 6: the full-order rule of Q2 elements, mf_spmv27 with two cells per wave; a preconditioner-side choice, "
                         "all fp64 -- the CG's operator, the residuals and the assembly keep the reference's 4, qf_cell(p+2)) or 4; the "
                         "default run reports the other one as config.with_smoother_quadrature_4")
    ap.add_argument("--fine-level", choices=["assembled", "matrix-free"], default="assembled",
                    help="assembled (default; the north-star path: global tangent scattered by colours + sell_spmv) or matrix-free: "
                         "no assembled fine tangent at all -- a tangent assembly writes point records, residual and the nodes' "
                         "diagonal blocks, every fine-level product runs on mf_spmv (tuning \"fine_level\" 1); the default run "
                         "reports it as config.with_matrix_free_fine_level over the same step window")
    ap.add_argument("--cg-start", choices=["zero", "previous-update", "previous-step", "extrapolated"], default="previous-step",
                    help="start vector of the linear solves: previous-step (default, what the executable sets: the j-th solve of "
                         "a step starts from the solution of the j-th solve of the previous step), extrapolated (the same over "
                         "two steps), zero, or previous-update as the reference's loop has it (nonlinear_elasticity.cc:419,472: "
                         "the previous Newton update of the step); the stopping rule is the same; the N = 1 line reports the "
                         "other choices as further measurements")
    ap.add_argument("--slabs", type=int, default=1, help="diagnostic: cut the mesh into this many slabs on ONE GPU")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N GPUs: strong = the one cells^3 block (BASELINE configuration 4) is cut into N parts (default); "
                         "weak = every GPU gets its own cells^3 block of a beam (cells x cells x N*cells)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the second (weak-scaling) measurement")
    ap.add_argument("--no-rccl", action="store_true",
                    help="diagnostic: with N ranks every rank solves its own copy of the single-GPU problem (no RCCL "
                         "communicator); exercises launcher, rendezvous and reporting on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-cells", type=int, default=34,
                    help="cells per side of the CPU-baseline sample (34 = BASELINE configuration 3; 0: skip)")
    ap.add_argument("--cpu-its", type=str, default="8,30", help="CG iterations timed by the CPU sample: SSOR,Jacobi")
    ap.add_argument("--no-cpu-config4", action="store_true",
                    help="skip the CPU leg on the metric's own mesh (BASELINE configuration 4, 59^3 cells: ONE Newton iteration = one "
                         "assembly + CG+Jacobi and CG+SSOR solves, ~5 minutes on a 64-core socket), which the default run includes")
    ap.add_argument("--cpu-config4", action="store_true", help=argparse.SUPPRESS)  # (round 3's opt-in; now the default)
    ap.add_argument("--cpu-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # child of pmc_traffic(): one step, no output
    ap.add_argument("--no-sides", action="store_true",
                    help="profiling runs: skip the further measurements beside the headline (config.with_*), so that a kernel trace "
                         "of the process holds the kernels of ONE path")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic passes (three child runs under rocprofv3 --pmc)")
    args = ap.parse_args()

    if args.cpu_worker:  # child of cpu_baseline(): CPU only
        c, a, b, full = (int(x) for x in args.cpu_worker.split(","))
        cpu_baseline_worker(c, a, b, full)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # nothing has touched the GPU in this process

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # The CPU leg runs FIRST, in child processes started while this process has not touched the GPU (no torch import, no
    # HIP context): a fork + exec out of a process that holds a HIP context is the fragile order.  It needs no GPU.
    cpu_leg = None
    if rank == 0 and world == 1 and args.cpu_cells > 0:
        its_a, its_b = (int(x) for x in args.cpu_its.split(","))
        budget = float(os.environ.get("MI_BENCH_CPU_BUDGET_S", "900"))
        cpu_leg = cpu_baseline(args.cpu_cells, its_a, its_b, budget, not args.no_cpu_config4 and args.cells == 59)

    # ... and so do the counter passes for `roofline.traffic`: children under rocprofv3, this process still GPU-free
    pmc, pmc_note = None, "not collected (N > 1, --slabs, --no-pmc or a child itself)"
    if rank == 0 and world == 1 and args.slabs == 1 and not args.no_pmc and not args.pmc_child and args.cells >= 24:
        pmc, pmc_note = pmc_traffic(args.cells, float(os.environ.get("MI_BENCH_PMC_BUDGET_S", "300")))

    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # one process per GPU; a launcher that narrows the visible devices to one per process is honoured
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py: no GPU visible (the hot path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    replicas = world > 1 and args.no_rccl
    if world > 1 and not replicas and ndev < world:
        print("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank; --no-rccl runs replicas)" % (world, ndev),
              file=sys.stderr)
        sys.exit(2)
    device = local_rank if local_rank < ndev else 0
    torch.cuda.set_device(device)
    M = _pkg()
    n = args.cells
    uid = None
    if world > 1:
        # control plane (rendezvous, unique-id broadcast, barriers, max over ranks) on gloo; the data path --
        # ghost-plane send/recv and scalar all-reduces of the CG -- runs on RCCL over xGMI inside the library
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        box = [M.comm_unique_id() if rank == 0 and not replicas else None]
        dist.broadcast_object_list(box, src=0)
        uid = None if replicas else box[0]
    parts = args.slabs if (world == 1 or replicas) else world
    traction = (0.0, -2e3, 0.0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, cells, steps, warmup, uid_, cg_start=None, cg_operator=None, smoother_precision=None, fine_level=None,
                diag_lag=None, quadrature=None):
        """K timed Newmark steps on the cells^3 block (strong) or the cells x cells x parts*cells beam (weak)"""
        nz = cells * parts if scaling == "weak" else cells
        G = M.Context(dim=3, degree=2, reps=(cells, cells, nz), lo=(0, 0, 0), hi=(1, 1, nz / cells), mu=0.5e6, nu=0.4,
                      rho=1000.0, beta=0.25, gamma=0.5, delta_t=0.005, device=device, rank=None if replicas else rank,
                      world=1 if replicas else world, unique_id=uid_, slabs=args.slabs if (world == 1 or replicas) else 1)
        G.set_tuning("smoother_operator", {"matrix-free": 2, "element": 1, "assembled": 0}[args.smoother_operator])
        G.set_tuning("cg_warm_start", {"zero": 0, "previous-update": 1, "previous-step": 2, "extrapolated": 3}[cg_start or args.cg_start])
        G.set_tuning("cg_operator", 1 if (cg_operator or args.cg_operator) == "element" else 0)
        G.set_tuning("precond", 1 if args.precond == "mg" else 0)
        if args.precond_storage == "f32":
            G.set_tuning("precond_storage", 32)
        if (smoother_precision or args.smoother_precision) == "f32":
            G.set_tuning("smoother_precision", 32)
        G.set_tuning("smoother_quadrature", quadrature or args.smoother_quadrature)
        if (fine_level or args.fine_level) == "matrix-free":
            G.set_tuning("fine_level", 1)
            G.set_tuning("mf_diag_lag", 0 if diag_lag == 0 else 1)  # (1: what the executable sets beside MI_FINE_LEVEL=1)

        def one_step(k):
            ramp = min(1.0, (k + 1) / 10.0)
            G.set_interface_traction(tuple(ramp * t for t in traction))
            _, info = G.newmark_step(tol_lin=args.tol_lin, max_it_mult=1.0)
            return info

        for k in range(warmup):
            one_step(k)
        G.set_profiling(True)
        G.reset_timings()
        barrier()
        t0 = time.perf_counter()
        newton = cg_its = assemblies = 0
        for k in range(steps):
            info = one_step(warmup + k)
            newton += info.newton_iterations
            cg_its += info.lin_its_total
            assemblies += info.assemblies
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [elapsed]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, elapsed)  # every rank's own clock over the window (control plane, gloo)
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        lin_its_last = [int(info.lin_its[i]) for i in range(min(info.newton_iterations, 16))]
        cnt = {k: G.get_tuning("count_" + k) for k in ("scalar_allreduce", "scalar_allreduce_cg", "vector_allreduce", "halo_exchange", "cg_host_sync",
                                                        "cg_iterations", "cg_solves", "mg_refresh")}
        cnt["mg_refresh_every"] = G.get_tuning("mg_refresh_every")
        r = {"G": G, "elapsed": elapsed, "lin_its_last": lin_its_last, "counts": cnt, "newton": newton, "cg_its": cg_its, "assemblies": assemblies, "nz": nz,
             "tm": G.timings(), "comm": G.comm_info(), "per_rank_s": per_rank}
        return r

    if args.pmc_child:  # one Newmark step of the headline configuration under the counters, nothing to report
        measure(args.scaling, n, 1, 0, None)
        return
    R = measure(args.scaling, n, args.steps, args.warmup, uid)
    G, elapsed, tm, nz = R["G"], R["elapsed"], R["tm"], R["nz"]
    out = None
    if rank == 0:
        nnzb = G.nnz // 9
        ms_step = 1e3 * elapsed / args.steps
        spmv_ms, spmv_n = tm["spmv"]
        spmv_avg_ms = spmv_ms / max(spmv_n, 1)
        share = 1 if replicas else world
        bytes_bsr = spmv_bytes(G.nnodes, nnzb, 3) // share  # bytes of the rows this rank owns
        achieved = bytes_bsr / (spmv_avg_ms * 1e-3) / 1e9 if spmv_n else 0.0
        # whole-step view: the algorithmic bytes of every fine-level product and assembly of a step / the step time
        out = {
            "metric": "DoF-updates/sec per Newmark step (assembly+CG), 3D Q2 ~5M DoFs",
            "value": G.n * (world if replicas else 1) * args.steps / elapsed,
            "unit": "DoF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "replicas" if replicas else args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "nonlinear_elasticity 3D Q2 neo-Hookean block %dx%dx%d cells, %d DoFs, %d nnz, "
                            "Newton+Newmark, %s-PCG Residual=%g, traction (0,-2e3,0) Pa ramped over 10 steps, dt=0.005; "
                            "boundary roles: face x- clamped (all components, boundary id 1 of the reference), the other FIVE "
                            "faces x+ y- y+ z- z+ are the coupling interface (id 7) and carry the traction -- no out-of-plane clamp "
                            "(id 8) on z+-: SURVEY 8d's 'choose and state'"
                            % (n, n, nz, G.n, G.nnz, "multigrid" if args.precond == "mg" else "Jacobi", args.tol_lin),
                "preconditioner": "geometric multigrid V-cycle (Chebyshev block-Jacobi smoothing, re-assembled coarse levels; Q2 "
                                  "and Q1 levels of the fine cells distributed over the slabs, coarser levels replicated)"
                if args.precond == "mg" else "Jacobi",
                "preconditioner_storage": args.precond_storage,
                "smoother_precision": args.smoother_precision,
                "smoother_quadrature": args.smoother_quadrature,
                "n_dofs": G.n,
                "nnz": G.nnz,
                "decomposition": ("single GPU" if args.slabs == 1 else "%d slabs cut along %s, emulated on one GPU"
                                  % (args.slabs, " xyz"[G.get_tuning("cut_axis")])) if world == 1 else
                ("%d independent replicas (--no-rccl diagnostic)" % world if replicas else
                 "%d slabs cut along %s (one per GPU; automatic choice: the direction with most cell layers, ties -> z), ghost-cell "
                 "redundant assembly, RCCL send/recv halo + all-reduce" % (world, " xyz"[G.get_tuning("cut_axis")])),
                "team_size": R["comm"][0],
                "rccl_ranks": R["comm"][1],  # ncclCommCount of the library's communicator (0: no RCCL in this run)
                "per_rank_ms_per_step": [1e3 * t / args.steps for t in R["per_rank_s"]],  # each rank's own clock; value uses the max
                "newton_iterations_per_step": R["newton"] / args.steps,
                "cg_iterations_per_step": R["cg_its"] / args.steps,
                "assemblies_per_step": R["assemblies"] / args.steps,
                "cg_iterations_last_step": R["lin_its_last"],
                "cg_start": {"zero": "zero for every solve (the reference starts the 2nd, 3rd ... solve of a step from the previous "
                                     "Newton update: same stopping rule, more iterations; --cg-start previous-update)",
                             "previous-update": "previous Newton update, as in the reference (nonlinear_elasticity.cc:419,472)",
                             "previous-step": "the j-th solve of a step starts from the solution of the j-th solve of the previous "
                                              "step (what the executable sets; same stopping rule; with_cg_start_zero and "
                                              "with_cg_start_previous_update are the runs with the other start vectors)",
                             "extrapolated": "as previous-step, extrapolated linearly over two steps"}[args.cg_start],
                # latency-bound events of the linear solves per CG iteration / per solve.  Complete for a decomposed run
                # (N ranks or --slabs N); on ONE slab the scalar all-reduce sites are still counted (they are no-ops there)
                # while halo exchanges and the V-cycle's vector all-reduce are skipped before their counters
                # scalar_allreduce counts every one of a step (the Newton loop's residual and update norms included, seven per
                # step), scalar_allreduce_in_solver those made inside the linear solves
                "cg_recurrence": "single reduction (r.z, z.Az, ||r||^2 in one all-reduce per iteration)" if G.get_tuning("cg_single_reduction_active") else "standard (p.Ap between the two updates)",
                "collectives_per_iteration": {
                    "scalar_allreduce": R["counts"]["scalar_allreduce"] / max(R["counts"]["cg_iterations"], 1),
                    "scalar_allreduce_in_solver": R["counts"]["scalar_allreduce_cg"] / max(R["counts"]["cg_iterations"], 1),
                    "vector_allreduce": R["counts"]["vector_allreduce"] / max(R["counts"]["cg_iterations"], 1),
                    "halo_exchange": R["counts"]["halo_exchange"] / max(R["counts"]["cg_iterations"], 1)},
                "host_syncs_per_solve": R["counts"]["cg_host_sync"] / max(R["counts"]["cg_solves"], 1),
                # the multigrid preconditioner's coarse operators (levels >= 1) are kept over time steps: rebuilt at every
                # k-th step, or earlier when a solve needs a quarter more iterations than the first one after a rebuild
                "coarse_operator_refresh": {"every_steps": R["counts"]["mg_refresh_every"],
                                            "refreshes_in_timed_steps": R["counts"]["mg_refresh"]},
                "ms_assembly_per_step": tm["assemble_total"][0] / args.steps,
                "ms_cg_per_step": tm["cg_total"][0] / args.steps,
                "ms_diagonal_blocks_per_step": tm["assemble_diag"][0] / args.steps,  # matrix-free fine level only
                "fine_level": args.fine_level,
                "ms_assemble_cells_per_assembly": tm["assemble_cells"][0] / max(tm["assemble_cells"][1], 1),
                "ms_assemble_residual_only_pass": tm["assemble_residual"][0] / max(tm["assemble_residual"][1], 1),
                "tangent_assemblies_per_step": tm["assemble_cells"][1] / args.steps,
                "residual_only_passes_per_step": tm["assemble_residual"][1] / args.steps,
            },
            "roofline": {
                "kernel": "sell_spmv<3,true,true,false,false,true> = <D=3, non-temporal matrix loads, DOT=true, fp64 values, no fused "
                          "smoother update, generated column indices>: the CG's q = K p with fused p.q partials on the assembled "
                          "tangent itself (x-line-interleaved block rows: the array the element scatter writes is the array the "
                          "product streams, 13.5 / 22.5 KiB chunks by LDS-DMA; no layout copy since round 3)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes from PMC counters cannot be collected from inside this process: the per-launch figure of the
                # committed counter passes is quoted under its own name, `traffic` itself stays null
                "traffic": None,
                "algorithmic_GB_per_launch": bytes_bsr / 1e9,
                "bytes_per_launch": bytes_bsr,
                "launches_timed": spmv_n,
                "avg_launch_ms": spmv_avg_ms,
                "timing": "start/stop events of the dispatch itself (hipExtLaunchKernelGGL), on the library's stream",
                "achieved_scalar_csr_equivalent": spmv_bytes_scalar_csr(G.nnodes, nnzb, 3) / share / (spmv_avg_ms * 1e-3) / 1e9
                if spmv_n else 0.0,
            },
        }
        # whole-step view next to the per-kernel fraction: algorithmic bytes of all fine-level products (CG + smoother),
        # tangent assemblies (8 nnz + connectivity + gathers + rhs), residual-only passes and layout copies of
        # one step / the step's wall time
        n_prod = (spmv_n + tm["spmv_precond"][1]) / args.steps
        form = G.get_tuning("smoother_operator_active")  # 2 matrix-free (records), 1 element tangents, 0 assembled
        ebe = form in (1, 2)
        single = False
        # per cell: element tangents = 378 lower-triangle 3x3 blocks, matrix-free = 11 doubles per quadrature point (64:
        # F, J^-2/3, 1/J; MF_NREC in mi_kernels.h); both + 27 node ids + first-touch bits per cell, x gathered, y written
        per_cell = 378 * 72 if form == 1 else 11 * 64 * 8
        ebe_bytes = G.ncells * (per_cell + 27 * 4 + 4) + 8 * G.n * 2
        asm_bytes = 8 * G.nnz + 4 * G.ncells * 27 + 16 * G.ncells * 81 + 8 * G.n + (G.ncells * per_cell if ebe else 0)
        res_bytes = 4 * G.ncells * 27 + 16 * G.ncells * 81 + 8 * G.n
        mf_level = args.fine_level == "matrix-free"
        if mf_level:
            # no tangent is written: a tangent pass = the point pass (gathers, records written, residual through the slots) ...
            asm_bytes = 16 * G.ncells * 81 + G.ncells * per_cell + 2 * 8 * G.ncells * 81 + 8 * G.n
            # ... + per diagonal-block pass: records read, the cells' slots written and read, the node data written
            diag_bytes = G.ncells * per_cell + 2 * 48 * G.ncells * 27 + (72 + 72 + 48 + 24) * G.nnodes
        cg_bytes = ebe_bytes if (mf_level or (args.cg_operator == "element" and ebe)) else spmv_bytes(G.nnodes, nnzb, 3)
        sm_bytes = ebe_bytes
        if form == 2 and G.get_tuning("smoother_quadrature_active") == 3:
            sm_bytes = G.ncells * (11 * 27 * 8 + 4) + 8 * G.n * 2
        step_bytes = (spmv_n / args.steps * cg_bytes + (tm["assemble_diag"][1] / args.steps * diag_bytes if mf_level else 0) +
                      tm["spmv_precond"][1] / args.steps * (sm_bytes if ebe else spmv_bytes(G.nnodes, nnzb, 3)) +
                      tm["assemble_cells"][1] / args.steps * asm_bytes +
                      tm["assemble_residual"][1] / args.steps * res_bytes +
                      120 * G.n) / share
        out["config"]["smoother_operator"] = (
            "matrix-free from the quadrature-point records of its own 27-point rule, rewritten per tangent (%.2f GB per product)"
            % (sm_bytes / 1e9) if form == 2 and sm_bytes != ebe_bytes
            else "matrix-free from the quadrature-point records of the assembly (%.2f GB per product)" % (ebe_bytes / 1e9) if form == 2
            else "unassembled symmetric element tangents (%.2f GB per product)" % (ebe_bytes / 1e9) if form == 1
            else "assembled sliced-ELL matrix")
        out["config"]["ms_smoother_fine_product"] = tm["spmv_precond"][0] / max(tm["spmv_precond"][1], 1)
        if (args.cg_operator == "element" or args.fine_level == "matrix-free") and ebe:
            # A/B run / matrix-free fine level: the CG's product is the unassembled form too; quote it on those bytes
            cgp = out["roofline"]
            cgp["kernel"] = ("the CG's q = K p as mf_spmv (one launch over all cells) + mf_gather_dot (slot sum with the partials of p.q): "
                             "the matrix-free fine level" if args.fine_level == "matrix-free" else
                             "the CG's q = K p in the smoother's unassembled form + a separate p.q reduction (A/B option)")
            cgp["bytes_per_launch"] = ebe_bytes
            cgp["algorithmic_GB_per_launch"] = ebe_bytes / 1e9
            cgp["achieved"] = ebe_bytes / (spmv_avg_ms * 1e-3) / 1e9 if spmv_n else 0.0
            cgp["frac"] = cgp["achieved"] / HBM_PEAK_GBS
            cgp.pop("achieved_scalar_csr_equivalent", None)
        if ebe and tm["ebe_launch"][1] > 0 and world == 1 and args.slabs == 1:
            # the DOMINANT kernel of the step (half of the GPU time) is the element-tangent product of the multigrid smoother:
            # the roofline object is quoted on it, the CG's product (the kernel north_star names) moves to `cg_product`
            cg = out["roofline"]
            ebe_ms = tm["ebe_launch"][0] / tm["ebe_launch"][1]
            per_launch = ebe_bytes / 8  # one colour of the eight; the colours differ by +-5 % in cells
            single = form == 2 and G.get_tuning("mf_single_launch") == 1
            q27 = single and G.get_tuning("smoother_quadrature_active") == 3
            if single:
                # ONE launch over all cells: records + node ids + slot ids per cell, x once, 81 contributions per cell
                # written to the cell's slots (the sum over the slots of a node is a second, streaming launch: mf_gather)
                per_launch = G.ncells * (11 * 64 * 8 + 27 * 4 + 27 * 4 + 81 * 8) + 8 * G.n
            if q27:
                # the smoother's 27-point form: 27 x 11 record numbers per cell, node ids by arithmetic, cell-major slots (no slot ids)
                per_launch = G.ncells * (11 * 27 * 8 + 81 * 8) + 8 * G.n
            out["roofline"] = {
                "kernel": ("mf_spmv27: the smoother's fine-level product P^T K_e P x of all cells, evaluated from 27 x 11 quadrature-point "
                           "numbers per cell (F, J^(-2/3), 1/J at the 3 x 3 x 3 Gauss points: the full-order rule of Q2 elements; the "
                           "CG's operator keeps the assembly's 64 points), sum factorised, TWO cells per wavefront (27 work items per "
                           "stage and cell); " if q27 else
                           "mf_spmv: the cells' P^T K_e P x evaluated from the 64 x 11 quadrature-"
                           "point numbers per cell the tangent is linearised at (sum factorisation, no stored K_e; 4.8x fewer bytes "
                           "than the element tangents it replaced, which takes the product off the HBM roofline: VALU, LDS and HBM "
                           "are each half to three quarters busy, profiles/r05/pmc_counters_mf_spmv_n59.json); "
                           if form == 2 else
                           "ebe_spmv: y += sum over the cells of ONE colour of P^T K_e P x with the unassembled symmetric element "
                           "tangents (378 lower-triangle 3x3 blocks per cell); ") +
                          ("one launch over all cells (every cell stores into its own slots) + mf_gather (sum over the slots of a "
                           "node, in the order of the colour-by-colour update) " if single else "eight launches ") +
                          "= one fine-level product of the multigrid smoother; the kernel with the largest share "
                          "of the step's GPU time",
                # a ridge kernel, not an HBM stream: it moves 4.8x fewer bytes than the element tangents it replaced and
                # sits at about half of BOTH the HBM and the FP64 vector roof (`fp64` below); `frac` stays the HBM fraction
                # of its algorithmic bytes, as the task defines it
                "bound": "hbm+valu" if form == 2 else "hbm",
                "achieved": per_launch / (ebe_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": per_launch / (ebe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_GB_per_launch": per_launch / 1e9,
                "bytes_per_launch": per_launch,
                "launches_timed": tm["ebe_launch"][1],
                "avg_launch_ms": ebe_ms,
                "timing": "start/stop events of the dispatch itself (hipExtLaunchKernelGGL) on every 6th product's launch(es)",
                "share_of_step": tm["spmv_precond"][0] / args.steps / ms_step,
                "cg_product": cg,
            }
            if form == 2 and single:
                # FP64 operations of one launch: per cell and lane-slot the kernel issues MF_FP64_OPS_PER_CELL_WAVE wave
                # instructions of FP64 arithmetic (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 of the committed counter pass, FMA = 2),
                # i.e. counted over all 64 lanes of the cell's wavefront whether a stage uses them or not
                flops = G.ncells * 64.0 * MF_FLOPS_PER_LANE
                src = "profiles/r05/pmc_counters_mf_spmv_n59.json (2 x FMA + MUL + ADD wave instructions x 64 lanes per cell)"
                if q27:
                    flops = ((G.ncells + 1) // 2) * 64.0 * MF27_FLOPS_PER_LANE
                    src = ("instruction count of the kernel's ISA: 267 FMA + 96 MUL + 12 ADD FP64 wave instructions per wavefront of two cells "
                           "x 64 lanes (profiles/r06/pmc_counters_mf_spmv27_n59.json)")
                out["roofline"]["fp64"] = {
                    "flops_per_launch": flops, "TFLOP_per_s": flops / (ebe_ms * 1e-3) / 1e12, "peak_TFLOP_per_s": FP64_VECTOR_PEAK_TF,
                    "frac": flops / (ebe_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF, "source": src}
        out["roofline"]["whole_step"] = {
            "fine_level_products_per_step": n_prod, "algorithmic_GB_per_step": step_bytes / 1e9,
            "GB_per_s": step_bytes / 1e9 / (ms_step * 1e-3), "frac": step_bytes / 1e9 / (ms_step * 1e-3) / HBM_PEAK_GBS,
            "ms_fine_products_precond_per_step": tm["spmv_precond"][0] / args.steps,
            "ms_fine_products_cg_per_step": spmv_ms / args.steps}
        if world == 1 and args.slabs == 1 and args.fine_level == "assembled":
            # streaming calibration in the SAME process: a pure read of the 8*nnz-byte block-CSR value array (the box's
            # achievable HBM ceiling; boxes of the pool differ on the gather-heavy product, not on this kernel)
            G.set_tuning("spmv_variant", 13)
            ms_cal = G.bench_spmv(10)
            G.set_tuning("spmv_variant", 3)
            out["roofline"]["calibration_stream_read"] = {"ms": ms_cal, "GB": 8 * G.nnz / 1e9,
                                                          "GB_per_s": 8 * G.nnz / ms_cal / 1e6}
        # `traffic`: HBM bytes per launch from the counter passes of THIS run (children started at the top of main)
        def live_traffic(prefix, min_bytes):
            rows = [v for k, v in (pmc or {}).items() if k.startswith(prefix) and v["bytes_per_launch"] >= min_bytes]
            if not rows:
                return None
            n_l = sum(v["launches"] for v in rows)
            return sum(v["bytes_per_launch"] * v["launches"] for v in rows) / n_l, n_l

        how_live = ("LIVE: `rocprofv3 --pmc <counter> -- python3 bench.py --pmc-child` (one Newmark step of this configuration), one "
                    "child process per counter before this process touched the GPU; TCC_EA0_RDREQ_sum x 128 B - "
                    "TCC_EA0_RDREQ_32B_sum x 96 B + WRITE_SIZE x 1 KiB (MI355X guide, gfx950 corrections); " + pmc_note)
        cgp = out["roofline"].get("cg_product", out["roofline"])
        dom = out["roofline"] if "cg_product" in out["roofline"] else None
        t_cg = live_traffic("mi::sell_spmv<3, true, true", 1e9) if args.cg_operator == "assembled" and args.fine_level == "assembled" else None
        if t_cg:
            cgp["traffic"], cgp["traffic_launches"] = t_cg
            cgp["traffic_ratio_to_algorithmic"] = t_cg[0] / cgp["bytes_per_launch"]
            cgp["traffic_source"] = how_live
        q27_dom = dom is not None and "mf_spmv27" in dom["kernel"]
        t_dom = live_traffic("mi::mf_spmv27" if q27_dom else "mi::mf_spmv<" if form == 2 else "mi::ebe_spmv", 0.3e9 if single else 0.0) if dom is not None else None
        if t_dom:
            dom["traffic"], dom["traffic_launches"] = t_dom
            dom["traffic_ratio_to_algorithmic"] = t_dom[0] / dom["bytes_per_launch"]
            dom["traffic_source"] = how_live
        elif dom is not None or not t_cg:
            out["roofline"]["traffic_note"] = "traffic is null: " + pmc_note
        t_asm = live_traffic("mi::assemble_q2sf<false", 0.0)
        if t_asm:
            out["roofline"]["assemble_q2sf_traffic"] = {"bytes_per_colour_launch": t_asm[0], "launches": t_asm[1],
                                                        "bytes_per_tangent_assembly": 8 * t_asm[0], "source": how_live}
        pmc_file = os.path.join(ROOT, "profiles", "r05", "pmc_bench_n59.json")
        if world == 1 and args.slabs == 1 and n == 59 and os.path.exists(pmc_file):
            # NOT a measurement of this run: per-launch HBM traffic of the same command under `rocprofv3 --pmc`
            # (tools/pmc_bench.sh), as committed with the round's profiles; `roofline.traffic` itself stays null because
            # PMC counters cannot be collected from inside this process
            pmc = json.load(open(pmc_file))
            how = ("rocprofv3 --pmc over `python3 bench.py --steps 1 --warmup 0 --cpu-cells 0`, one counter per pass: "
                   "TCC_EA0_RDREQ x 128 B - TCC_EA0_RDREQ_32B x 96 B + WRITE_SIZE x 1 KiB (tools/pmc_bench.sh)")
            ref = {"note": "committed counter passes of another process of the same command, not of this run", "how": how,
                   "source": os.path.relpath(pmc_file, ROOT)}
            dot = [v for k, v in pmc.items() if k.startswith("mi::sell_spmv<3, true, true")]
            if dot:
                ref["cg_product"] = {"GB_per_launch": dot[0]["traffic_GB_per_launch"],
                                     "ratio_to_algorithmic": dot[0]["traffic_GB_per_launch"] / (bytes_bsr / 1e9)}
            eb = [v for k, v in pmc.items() if k.startswith("mi::mf_spmv" if form == 2 else "mi::ebe_spmv")
                  and (not single or v["traffic_GB_per_launch"] > 1.0)]
            if eb and "cg_product" in out["roofline"]:
                tot = sum(v["traffic_GB_per_launch"] * v["launches"] for v in eb) / sum(v["launches"] for v in eb)
                ref["dominant_kernel"] = {"GB_per_launch": tot,
                                          "ratio_to_algorithmic": tot / out["roofline"]["algorithmic_GB_per_launch"]}
            asm = [v for k, v in pmc.items() if k.startswith("mi::assemble_q2sf<false")]
            if asm:
                ref["assemble_q2sf_per_tangent_assembly"] = {
                    "GB": sum(v["traffic_GB_per_launch"] * v["launches"] for v in asm) / (sum(v["launches"] for v in asm) / 8.0),
                    "note": "eight colour launches per assembly (launch-weighted mean per launch x 8)"}
            out["roofline"]["reference_profiles"] = ref
    del G, R
    # ---- second field for N > 1: weak scaling (one cells^3 block per GPU)
    if world > 1 and not replicas and not args.no_weak and args.scaling == "strong":
        uid2 = None
        box = [M.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid2 = box[0]
        W = measure("weak", n, args.steps, args.warmup, uid2)
        if rank == 0:
            out["weak_scaling"] = {"value": W["G"].n * args.steps / W["elapsed"], "unit": "DoF-updates/s",
                                   "ms_per_step": 1e3 * W["elapsed"] / args.steps, "n_dofs": W["G"].n,
                                   "workload": "%dx%dx%d cells (%d^3 per GPU)" % (n, n, W["nz"], n),
                                   "cg_iterations_per_step": W["cg_its"] / args.steps}
        del W
    if world == 1 and args.slabs == 1 and not args.no_sides:
        # Further measurements beside the headline, each over the SAME step window as the headline (the same warm-up
        # steps, the same ramp phase, the same coarse-operator refreshes inside the window): `steps` / `warmup` in every
        # sub-object say so.  (Until round 5 these ran 3 steps after 1 warm-up and were indicative only.)
        def side(**kw):
            S = measure(args.scaling, n, args.steps, args.warmup, None, **kw)
            r = {"ms_per_step": 1e3 * S["elapsed"] / args.steps, "value": S["G"].n * args.steps / S["elapsed"],
                 "cg_iterations_per_step": S["cg_its"] / args.steps, "cg_iterations_last_step": S["lin_its_last"],
                 "newton_iterations_per_step": S["newton"] / args.steps, "steps": args.steps, "warmup": args.warmup,
                 "window": "the headline's: steps %d..%d of the ramp" % (args.warmup + 1, args.warmup + args.steps),
                 "ms_assembly_per_step": S["tm"]["assemble_total"][0] / args.steps, "ms_cg_per_step": S["tm"]["cg_total"][0] / args.steps}
            tmS = S["tm"]
            del S
            return r, tmS

        # the other start vectors: zero for every solve, and the reference's (the previous Newton update,
        # nonlinear_elasticity.cc:419,472)
        for other in ("zero", "previous-update", "previous-step"):
            if other == args.cg_start or (other == "previous-step" and args.cg_start == "extrapolated"):
                continue
            out["config"]["with_cg_start_" + other.replace("-", "_")], _ = side(cg_start=other)
        if args.smoother_precision == "f64" and args.smoother_operator == "matrix-free" and n >= 24:
            # opt-in A/B beside the headline: the smoother's matrix-free products in fp32 (preconditioner-only change)
            out["config"]["with_smoother_precision_f32"], _ = side(smoother_precision="f32")
            out["config"]["with_smoother_precision_f32"]["note"] = (
                "opt-in, not the headline: fp32 arithmetic and records in the smoother's fine-level products only -- on the 64-point "
                "kernel (mf_spmv<..., float>), i.e. to be compared with with_smoother_quadrature_4, not with the 27-point headline")
        if args.cg_operator == "assembled" and args.fine_level == "assembled" and n >= 24:
            # opt-in A/B beside the headline: the CG's own product on the element tangents too (no sliced-ELL copy);
            # not the default because north_star names the product on the assembled matrix
            out["config"]["with_cg_operator_element"], _ = side(cg_operator="element")
        if args.fine_level == "assembled" and n >= 8:
            # Round 6: the fine level matrix-free END TO END -- no assembled fine tangent: a tangent assembly = residual pass
            # that also writes the point records + the nodes' diagonal blocks from them (mf_diag); the CG's product, residual
            # and start-vector products and the smoother all on mf_spmv.  Same converged steps (oracle tests parametrised
            # over it).  `value` stays on the north-star path (global CSR + sell_spmv); this is the second number.
            r, tmS = side(fine_level="matrix-free")
            r["ms_tangent_pass_records_residual"] = tmS["assemble_cells"][0] / max(tmS["assemble_cells"][1], 1)
            r["ms_diagonal_blocks"] = tmS["assemble_diag"][0] / max(tmS["assemble_diag"][1], 1)
            r["ms_cg_product"] = tmS["spmv"][0] / max(tmS["spmv"][1], 1)
            r["tangent_assemblies_per_step"] = tmS["assemble_cells"][1] / args.steps
            r["diagonal_blocks"] = ("formed at the first tangent of a time step and kept over its Newton iterations (\"mf_diag_lag\" 1, a "
                                    "preconditioner-side policy like the coarse operators' lag); every tangent: see "
                                    "with_matrix_free_fine_level_diagonal_every_tangent")
            r2, tm2 = side(fine_level="matrix-free", diag_lag=0)
            r2["ms_diagonal_blocks"] = tm2["assemble_diag"][0] / max(tm2["assemble_diag"][1], 1)
            r2["diagonal_block_passes_per_step"] = tm2["assemble_diag"][1] / args.steps
            out["config"]["with_matrix_free_fine_level_diagonal_every_tangent"] = r2
            r["diagonal_block_passes_per_step"] = tmS["assemble_diag"][1] / args.steps
            r["note"] = ("tuning \"fine_level\" 1 (bench.py --fine-level matrix-free): no global fine tangent is assembled or "
                         "stored (7.6 GB at 5 M DoFs released); not the headline because north_star names the CSR + SpMV path")
            out["config"]["with_matrix_free_fine_level"] = r
        if args.smoother_quadrature == 3 and args.smoother_operator == "matrix-free" and n >= 8:
            # The smoother's fine-level operator with the assembly's 4 x 4 x 4 Gauss points instead of the 3 x 3 x 3 the library
            # defaults to since round 6 (mf_spmv27: two cells per wave from 27-point records; preconditioner side only, all fp64:
            # the CG's operator, every residual and the assembly keep the reference's rule).  On both fine levels, over the
            # headline's step window: what the headline and with_matrix_free_fine_level would be with the rounds-3-to-5 smoother.
            note = ("tuning \"smoother_quadrature\" 4: the V-cycle's fine-level smoother products, residual and eigenvalue estimate with "
                    "the 64-point rule of the assembly (rounds 3-5); the default is 3 (27 points): same Newton tables and CG iterations")
            r3, tm3 = side(quadrature=4)
            r3["ms_smoother_fine_product"] = tm3["spmv_precond"][0] / max(tm3["spmv_precond"][1], 1)
            r3["note"] = note
            out["config"]["with_smoother_quadrature_4"] = r3
            if args.fine_level == "assembled":
                r4, tm4 = side(quadrature=4, fine_level="matrix-free")
                r4["ms_smoother_fine_product"] = tm4["spmv_precond"][0] / max(tm4["spmv_precond"][1], 1)
                r4["note"] = note
                out["config"]["with_matrix_free_fine_level_smoother_quadrature_4"] = r4
    if rank == 0 and world == 1 and args.cpu_cells > 0:
        # the GPU on the CPU sample's own configuration, beside it
        if args.cpu_cells != n or args.slabs != 1:
            S = measure("strong", args.cpu_cells, 3, 1, None)
            gpu_same = {"value": S["G"].n * 3 / S["elapsed"], "ms_per_step": 1e3 * S["elapsed"] / 3, "n_dofs": S["G"].n,
                        "cg_iterations_per_step": S["cg_its"] / 3, "steps": 3, "warmup": 1,
                        "window": "steps 2..4 of the ramp (the CPU leg times the first step of the same mesh)"}
            del S
        else:
            gpu_same = {"value": out["value"], "ms_per_step": out["ms_per_step"], "n_dofs": out["config"]["n_dofs"]}
        out["cpu_baseline"] = cpu_leg  # measured before the first GPU call of this process (top of main)
        out["cpu_baseline"]["gpu_same_config"] = gpu_same
        c4 = out["cpu_baseline"].get("config4_one_newton_iteration")
        if c4 and c4.get("live") and "t_assembly_s" in c4:
            # the GPU's first Newton iteration of a step on the same mesh: one tangent assembly + the first solve's share
            c4["gpu_first_newton_iteration_ms_estimate"] = (
                out["config"]["ms_assemble_cells_per_assembly"] +
                out["config"]["ms_cg_per_step"] / max(out["config"]["newton_iterations_per_step"], 1))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
