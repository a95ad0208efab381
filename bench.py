#!/usr/bin/env python3
"""bench.py -- DoF-updates/s of one Newmark step (Newton x (assembly + CG) + Newmark updates) of the
neo-Hookean solver on a synthetic 3D Q2 block (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one full Newmark step on the resident mesh under the ramped constant interface traction
(0,-2e3,0) Pa (ramp over the first 10 steps).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


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
        generated_columns = os.environ.get("MI_SELL_ICOL", "1") != "0"
    n = nnodes * dim
    index_bytes = 8 * nnodes if generated_columns else 4 * nnzb
    return 8 * nnzb * dim * dim + index_bytes + 4 * (nnodes + 1) + 8 * n + 8 * n


def spmv_bytes_scalar_csr(nnodes, nnzb, dim):
    """SURVEY.md section 8(d) figure for a scalar CSR of the same matrix: 12*nnz + 4*(N+1) + 16*N"""
    n = nnodes * dim
    return 12 * nnzb * dim * dim + 4 * (n + 1) + 16 * n


def cpu_baseline(n_cells_side, threads):
    """one Newmark step of the CPU oracle (restatement of the reference algorithm: WorkStream-style
    assembly over all host threads, CG + SSOR(0.65)) on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.lib().orc_set_threads(threads)
    d = O.make_desc(dim=3, degree=2, reps=(n_cells_side,) * 3)
    P = O.Problem(d)
    P.set_interface_traction((0.0, -2e2, 0.0))  # first ramp step
    t0 = time.perf_counter()
    rc, info = P.newmark_step(O.SOLVER_CG_SSOR, tol_lin=1e-6, max_it_mult=1.0)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {
        "value": P.n / dt,
        "unit": "DoF-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": "1 Newmark step, 3D Q2 block %d^3 cells (%d DoFs), CG+SSOR(0.65) tol 1e-6, %d Newton its, "
                  "%d CG its, assembly %.1fs + solve %.1fs; restatement of the reference algorithm, not the "
                  "deal.II binary" % (n_cells_side, P.n, info.newton_iterations, info.lin_its_total,
                                      info.t_assemble, info.t_solve),
    }


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
    ap.add_argument("--slabs", type=int, default=1, help="diagnostic: cut the mesh into this many slabs on ONE GPU")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N GPUs: weak = every GPU gets its own cells^3 block of the beam (cells x cells x N*cells, default); "
                         "strong = the one cells^3 block is cut into N slabs")
    ap.add_argument("--no-rccl", action="store_true",
                    help="diagnostic: with N ranks every rank solves its own copy of the single-GPU problem (no RCCL "
                         "communicator); exercises launcher, rendezvous and reporting on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-cells", type=int, default=16, help="cells per side of the CPU-baseline sample (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    # one process per GPU; a launcher that narrows the visible devices to one per process is honoured
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py: no GPU visible (the hot path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    device = local_rank if local_rank < ndev else 0
    torch.cuda.set_device(device)
    M = _pkg()
    n = args.cells
    uid = None
    replicas = world > 1 and args.no_rccl
    if world > 1:
        # control plane (rendezvous, unique-id broadcast, barriers, max over ranks) on gloo; the data path --
        # ghost-plane send/recv and scalar all-reduces of the CG -- runs on RCCL over xGMI inside the library
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        box = [M.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = None if replicas else box[0]
    # z-slabs, one per GPU.  weak scaling: the block grows along z with the number of GPUs (cubic cells of the same
    # size, cells^3 of them per GPU: at N=1 exactly the 59^3 case the metric names); strong: the 59^3 block is cut
    parts = args.slabs if (world == 1 or replicas) else world
    nz = n * parts if args.scaling == "weak" else n
    G = M.Context(dim=3, degree=2, reps=(n, n, nz), lo=(0, 0, 0), hi=(1, 1, nz / n), mu=0.5e6, nu=0.4, rho=1000.0,
                  beta=0.25, gamma=0.5, delta_t=0.005, device=device, rank=None if replicas else rank,
                  world=1 if replicas else world, unique_id=uid, slabs=args.slabs if (world == 1 or replicas) else 1)
    G.set_tuning("precond", 1 if args.precond == "mg" else 0)
    if args.precond_storage == "f32":
        G.set_tuning("precond_storage", 32)
    nnzb = G.nnz // 9
    traction = (0.0, -2e3, 0.0)

    def one_step(k):
        ramp = min(1.0, (k + 1) / 10.0)
        G.set_interface_traction(tuple(ramp * t for t in traction))
        _, info = G.newmark_step(tol_lin=args.tol_lin, max_it_mult=1.0)
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    G.set_profiling(True)
    G.reset_timings()
    barrier()
    t0 = time.perf_counter()
    newton = cg_its = assemblies = 0
    for k in range(args.steps):
        info = one_step(args.warmup + k)
        newton += info.newton_iterations
        cg_its += info.lin_its_total
        assemblies += info.assemblies
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = G.timings()

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        spmv_ms, spmv_n = tm["spmv"]
        spmv_avg_ms = spmv_ms / max(spmv_n, 1)
        share = 1 if replicas else world
        bytes_bsr = spmv_bytes(G.nnodes, nnzb, 3) // share  # bytes of the rows this rank owns
        achieved = bytes_bsr / (spmv_avg_ms * 1e-3) / 1e9 if spmv_n else 0.0
        # HBM traffic of the same kernel on the same workload from the committed PMC passes (rocprofv3 --pmc cannot be
        # collected from inside this process); only quoted when the workload is the one that was profiled
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01", "pmc_spmv_icol_n59.json")
        if world == 1 and args.slabs == 1 and n == 59 and os.path.exists(pmc_file):
            traffic = json.load(open(pmc_file))["traffic_bytes_per_launch"] / 1e9  # GB per launch
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
                "workload": "nonlinear_elasticity 3D Q2 neo-Hookean block %dx%dx%d cells (%d^3 per GPU), %d DoFs, %d nnz, "
                            "Newton+Newmark, %s-PCG Residual=%g, traction (0,-2e3,0) Pa ramped over 10 steps, dt=0.005"
                            % (n, n, nz, n, G.n, G.nnz, "multigrid" if args.precond == "mg" else "Jacobi", args.tol_lin),
                "preconditioner": "geometric multigrid V-cycle (Chebyshev-Jacobi smoothing, re-assembled coarse levels; Q2 and "
                                  "Q1 levels of the fine cells distributed over the slabs, coarser levels replicated)"
                if args.precond == "mg" else "Jacobi",
                "preconditioner_storage": args.precond_storage,
                "n_dofs": G.n,
                "nnz": G.nnz,
                "decomposition": ("single GPU" if args.slabs == 1 else "%d slabs emulated on one GPU" % args.slabs) if world == 1 else
                ("%d independent replicas (--no-rccl diagnostic)" % world if replicas else
                 "%d z-slabs (one per GPU), ghost-cell redundant assembly, RCCL send/recv halo + all-reduce" % world),
                "newton_iterations_per_step": newton / args.steps,
                "cg_iterations_per_step": cg_its / args.steps,
                "assemblies_per_step": assemblies / args.steps,
                "ms_assembly_per_step": tm["assemble_total"][0] / args.steps,
                "ms_cg_per_step": tm["cg_total"][0] / args.steps,
                "ms_sell_copy_per_step": tm["sell_copy"][0] / args.steps,
                "ms_assemble_cells_per_assembly": tm["assemble_cells"][0] / max(tm["assemble_cells"][1], 1),
            },
            "roofline": {
                "kernel": "sell_spmv<3,2,0,1,true,false,false,true> = <D=3, 2 blocks in flight, no ablation, non-temporal matrix "
                          "loads, DOT=true, fp64 values, no fused smoother update, generated column indices>: the CG's q = K p with fused p.q partials on "
                          "the sliced-ELL copy of the block-CSR tangent; the preconditioner's products run DOT=false "
                          "instantiations of the same kernel",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "GB per launch (PMC: TCC_EA0_RDREQ x 128 B + WRITE_SIZE x 1 KiB, profiles/r01/pmc_spmv_icol_n59.json)",
                "algorithmic_GB_per_launch": bytes_bsr / 1e9,
                "bytes_per_launch": bytes_bsr,
                "launches_timed": spmv_n,
                "avg_launch_ms": spmv_avg_ms,
                "achieved_scalar_csr_equivalent": spmv_bytes_scalar_csr(G.nnodes, nnzb, 3) / share / (spmv_avg_ms * 1e-3) / 1e9
                if spmv_n else 0.0,
            },
        }
        if args.cpu_cells > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_cells, os.cpu_count() or 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
