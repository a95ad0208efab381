"""The driver-facing contract of bench.py and __graft_entry__.py, exercised on a small mesh: one JSON line with the
agreed keys, roofline and cpu_baseline objects, sane values; smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu


def _run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly ONE JSON line
    return json.loads(lines[0])


@gpu
def test_bench_json_line_contract():
    d = _run(["--cells", "8", "--steps", "2", "--warmup", "1", "--cpu-cells", "3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "DoF-updates/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["n_dofs"]
    assert n == 3 * 17 ** 3
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None  # the PMC passes were collected for the 59^3 workload only
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "gpu_same_config"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0
    assert "pinned" in c["sample"] and c["gpu_same_config"]["n_dofs"] == 3 * 7 ** 3
    w = r["whole_step"]
    assert w["fine_level_products_per_step"] >= d["config"]["cg_iterations_per_step"] and 0 < w["frac"] < 1
    assert d["scaling"] == "strong" and d["config"]["rccl_ranks"] == 0 and d["config"]["team_size"] == 1
    # round 6: the further measurements run over the headline's own step window and say so; the fine level matrix-free end
    # to end is one of them (same Newton / CG iteration counts within one per step)
    for k, v in d["config"].items():
        if k.startswith("with_"):
            assert (v["steps"], v["warmup"]) == (d["steps"], d["warmup"]) and "window" in v, k
    mf = d["config"]["with_matrix_free_fine_level"]
    assert mf["newton_iterations_per_step"] == d["config"]["newton_iterations_per_step"]
    assert abs(mf["cg_iterations_per_step"] - d["config"]["cg_iterations_per_step"]) <= 1.0
    assert mf["diagonal_block_passes_per_step"] == 1.0 and mf["tangent_assemblies_per_step"] >= 1.0
    assert d["config"]["with_matrix_free_fine_level_diagonal_every_tangent"]["diagonal_block_passes_per_step"] == mf["tangent_assemblies_per_step"]
    assert d["config"]["fine_level"] == "assembled" and d["config"]["per_rank_ms_per_step"] == [d["ms_per_step"]]


@gpu
def test_bench_gpus_2_as_a_bare_command():
    """`python bench.py --gpus N` is what the driver runs: with no WORLD_SIZE in the environment bench.py starts the
    N ranks itself (child processes, before anything touches the GPU) and relays rank 0's single JSON line.  On the
    one-GPU box the ranks are replicas (--no-rccl); without that switch it refuses loudly instead of hanging in RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-rccl", "--cells", "6", "--steps", "2",
           "--warmup", "1"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "replicas" and d["config"]["rccl_ranks"] == 0
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd[:4] + cmd[5:], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode != 0 and "visible GPUs" in out.stderr


@gpu
def test_bench_options_and_smoke():
    d = _run(["--cells", "6", "--steps", "1", "--warmup", "1", "--cpu-cells", "0", "--slabs", "2", "--scaling", "strong",
              "--precond", "jacobi"])
    assert "cpu_baseline" not in d and d["scaling"] == "strong" and "2 slabs" in d["config"]["decomposition"]
    d = _run(["--cells", "6", "--steps", "1", "--warmup", "1", "--cpu-cells", "0", "--slabs", "2", "--precond-storage", "f32",
              "--scaling", "weak"])
    assert d["scaling"] == "weak" and d["config"]["n_dofs"] == 3 * 13 * 13 * 25
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stderr[-2000:]


@gpu
def test_bench_under_the_launcher_two_ranks_control_plane():
    """the driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.  A
    one-GPU box cannot host two RCCL ranks, so `--no-rccl` lets both ranks run their own copy of the problem: this
    exercises RANK/LOCAL_RANK/WORLD_SIZE handling, the device choice when fewer GPUs than ranks are visible, the
    gloo rendezvous, the unique-id broadcast, barriers, max-over-ranks timing and rank-0-only reporting"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--cells", "6", "--no-rccl"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "replicas" and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * d["config"]["n_dofs"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9


def test_launch_ranks_has_a_wall_clock_limit():
    """verdict r05: a hung RCCL rendezvous on the first real multi-GPU run must not hang the bench.  `python bench.py --gpus N`
    starts its ranks as a process group of their own and waits MI_BENCH_RANKS_TIMEOUT_S for them; past the limit the group is
    ended and the bench exits 124.  (Runs without a GPU: with a limit of half a second the ranks are still importing.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MI_BENCH_RANKS_TIMEOUT_S"] = "0.5"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-rccl", "--cells", "6", "--steps", "1",
                          "--warmup", "0", "--cpu-cells", "0"], cwd=ROOT, capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 124, (out.returncode, out.stderr[-500:])
    assert "did not finish within" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_committed_profile_agrees_with_its_bench_line():
    """profiles/r06: the rocprofv3 --kernel-trace --stats summary and the bench line of the SAME process (bench.py under the
    profiler) name the same average duration for the roofline's kernel and for the CG product -- the profiler times every
    launch, the bench every 6th product from the dispatch's own events; and the driver-style line of that box is the one
    the documents quote.  CPU test: it reads committed files only."""
    import csv

    P = os.path.join(ROOT, "profiles", "r06")
    for tag, cg_kernel in (("headline", "mi::sell_spmv<3, true, true"), ("matrix_free_fine_level", "mi::mf_spmv<true, true, true")):
        stats = {}
        with open(os.path.join(P, "kernel_stats_bench_%s_n59.csv" % tag), newline="") as f:
            for row in csv.DictReader(f):
                stats[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) * 1e-6)
        line = json.load(open(os.path.join(P, "bench_under_rocprof_%s_n59.json" % tag)))
        r = line["roofline"]
        calls, avg_ms = next(v for k, v in stats.items() if "mi::mf_spmv27<true, true>" in k)
        assert r["kernel"].startswith("mf_spmv27") and calls > 100
        assert abs(r["avg_launch_ms"] - avg_ms) / avg_ms < 0.05, (tag, r["avg_launch_ms"], avg_ms)
        calls, avg_ms = next(v for k, v in stats.items() if cg_kernel in k)
        cg_ms = r["cg_product"]["avg_launch_ms"]
        if tag == "matrix_free_fine_level":  # (the bench times the product with its gather; the trace lists the two kernels)
            avg_ms += next(v for k, v in stats.items() if "mf_gather_dot" in k)[1]
        assert abs(cg_ms - avg_ms) / avg_ms < 0.05, (tag, cg_ms, avg_ms)
    plain = json.load(open(os.path.join(P, "bench_plain_same_box_n59.json")))
    assert plain["metric"] == line["metric"] and plain["steps"] == 20 and plain["warmup"] == 5 and plain["n_gpus"] == 1
    assert plain["roofline"]["traffic"] and 1.0 < plain["roofline"]["traffic_ratio_to_algorithmic"] < 1.5
    assert plain["cpu_baseline"]["value"] > 0 and plain["config"]["with_matrix_free_fine_level"]["ms_per_step"] <= 90.0
    readme = open(os.path.join(P, "README.md")).read()
    assert "%.2f ms per step" % plain["ms_per_step"] in readme


def test_profile_digest_reads_the_committed_set():
    """tools/r6_profile_digest.py prints the numbers profiles/r06/README.md quotes from the committed files (CPU test)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r6_profile_digest.py")], cwd=ROOT, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    plain = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_plain_same_box_n59.json")))
    assert ("headline: %.2f ms per step" % plain["ms_per_step"]) in out.stdout
    assert "with_matrix_free_fine_level" in out.stdout and "emulated slabs" in out.stdout and "rank_share_n59.txt" in out.stdout
