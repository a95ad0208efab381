"""GPU tests of the drop-in boundary: the `elasticity` executables (parameters.prm + replayed coupling partner)
against the CPU oracle driven through the same coupling script, and the linear model through the C-ABI.

Interface displacements are compared per completed window, matched by vertex coordinate order (both sides use
ascending x-dof order, adapter.h:313-321).  Tolerance 1e-8 relative (linear tolerance tightened to 1e-12).
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

M = load_pkg()
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-adapter_amd", "host")
CASES = os.path.join(ROOT, "tests", "cases")
TOL = 1e-8


def _run_case(name, exe, tmp_path, env=None):
    for f in ("parameters.prm", "precice-config.xml"):
        (tmp_path / f).write_text(open(os.path.join(CASES, name, f)).read())
    out = subprocess.run([os.path.join(HOST, exe)], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env) if env else None)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rows = [np.array(l.split(), dtype=float) for l in open(tmp_path / "displacement.log") if not l.startswith("#")]
    return out.stdout, rows


def _prm(name):
    txt = open(os.path.join(CASES, name, "parameters.prm")).read()

    def get(key):
        return re.search(r"set\s+" + re.escape(key) + r"\s*=\s*(.+)", txt).group(1).strip()

    return get


def _scenario_desc(get, dim, **kw):
    sc = get("Scenario")
    common = dict(degree=int(get("Polynomial degree")), mu=float(get("Shear modulus")), nu=float(get("Poisson's ratio")),
                  rho=float(get("rho")), delta_t=float(get("Time step size")),
                  body_force=tuple(float(x) for x in get("body forces").split(",")), **kw)
    if sc in ("FSI3", "PF"):
        return O.scenario_desc(sc, dim, **common)
    reps = tuple(int(x) for x in get("Repetitions").split(","))
    lo = tuple(float(x) for x in get("Lower corner").split(","))
    hi = tuple(float(x) for x in get("Upper corner").split(","))
    return O.make_desc(dim=dim, reps=reps[:dim], lo=lo[:dim], hi=hi[:dim], **common)


def _check_rows(rows, expected, dim):
    assert len(rows) == len(expected)
    for r, (t, u) in zip(rows, expected):
        assert abs(r[0] - t) < 1e-12
        got = r[1:].reshape(-1, dim)
        assert got.shape == u.shape
        assert np.abs(got - u).max() / np.abs(u).max() < TOL


def test_executable_nonlinear_explicit_2d(tmp_path):
    name = "fsi3_neo_2d_explicit"
    stdout, rows = _run_case(name, "elasticity", tmp_path)
    get = _prm(name)
    P = O.Problem(_scenario_desc(get, 2))
    ids = P.interface_nodes
    dt, exp = float(get("Time step size")), []
    for k in range(4):
        P.set_interface_traction((0.0, -40.0 * min(1.0, (k + 1) / 4.0)))
        rc, info = P.newmark_step(O.SOLVER_CG_SSOR, tol_lin=1e-12, max_it_mult=2.0)
        assert rc == 0
        exp.append(((k + 1) * dt, P.vec(O.V_U).reshape(-1, 2)[ids].copy()))
    _check_rows(rows, exp, 2)
    # console contract: banner, mesh statistics, Newton table, timer sections
    assert "Number of coupling nodes:" in stdout and "Number of degrees of freedom: %d" % P.n in stdout
    assert "SOLVER STEP" in stdout and "CONVERGED!" in stdout and "Timestep 4 @" in stdout
    for section in ("Setup system", "Assemble linear system", "Linear solver", "Advance adapter", "Output results"):
        assert section in stdout
    assert os.path.exists(tmp_path / "out" / "solution-000.vtk")
    # 2D: one VTK_LAGRANGE_QUADRILATERAL (70) per cell with (p+1)^2 points (nonlinear_elasticity.cc:1222-1225)
    vtk = open(tmp_path / "out" / "solution-000.vtk").read().split("\n")
    p1 = int(get("Polynomial degree")) + 1
    k = next(i for i, l in enumerate(vtk) if l.startswith("CELL_TYPES"))
    assert int(vtk[k].split()[1]) == P.ncells and set(vtk[k + 1:k + 1 + P.ncells]) == {"70"}
    k = next(i for i, l in enumerate(vtk) if l.startswith("CELLS"))
    assert all(int(l.split()[0]) == p1 * p1 and len(l.split()) == p1 * p1 + 1 for l in vtk[k + 1:k + 1 + P.ncells])
    # machine-readable step log next to the Newton table
    import json
    steps = [json.loads(l) for l in open(tmp_path / "out" / "steps.jsonl")]
    assert [s["timestep"] for s in steps] == [1, 2, 3, 4] and all(s["newton_iterations"] >= 1 for s in steps)
    assert steps[0]["n_dofs"] == P.n and all(s["linear_iterations"] >= s["newton_iterations"] for s in steps)


def test_executable_replays_a_per_vertex_force_trace(tmp_path):
    """BASELINE configuration 5: the FSI3 flap coupled to a replayed fluid-force trace that varies along the interface
    and in time (rows "t vertex fx fy", frames coarser than the windows -> per-vertex interpolation in time), through the
    executable; the oracle receives the same per-vertex tractions"""
    name = "fsi3_neo_2d_vertex_trace"
    get = _prm(name)
    P = O.Problem(_scenario_desc(get, 2))
    ids = P.interface_nodes
    xy = P.coords[ids]
    s = (xy[:, 0] - xy[:, 0].min()) / (xy[:, 0].max() - xy[:, 0].min())
    dt = float(get("Time step size"))
    frames_t = np.array([1, 3, 5]) * dt

    def force(t):
        return np.stack([8.0 * np.sin(np.pi * s) * np.cos(40.0 * t), -60.0 * s * (1.0 + 0.3 * np.sin(90.0 * t))], axis=1)

    frames = [force(t) for t in frames_t]
    with open(tmp_path / "fluid-forces.txt", "w") as f:
        f.write("# t vertex fx fy\n")
        for t, fr in zip(frames_t, frames):
            for v, row in enumerate(fr):
                f.write("%.17g %d %.17g %.17g\n" % (t, v, row[0], row[1]))
    _, rows = _run_case(name, "elasticity", tmp_path)
    # the coupling-mesh vertices the executable reports are the oracle's interface nodes, in the same order
    vtx = np.array([l.split() for l in open(tmp_path / "solid-vertices.txt") if not l.startswith("#")], dtype=float)
    assert np.array_equal(vtx[:, 0], np.arange(len(ids))) and np.abs(vtx[:, 1:] - xy).max() < 1e-15
    exp = []
    for k in range(5):
        t = (k + 1) * dt
        j = min(np.searchsorted(frames_t, t - 1e-12), len(frames_t) - 1)
        if j == 0 or abs(frames_t[j] - t) < 1e-12:
            tr = frames[j]
        else:
            w = (t - frames_t[j - 1]) / (frames_t[j] - frames_t[j - 1])
            tr = (1 - w) * frames[j - 1] + w * frames[j]
        P.set_interface_traction(tr)
        rc, _ = P.newmark_step(O.SOLVER_CG_SSOR, tol_lin=1e-12, max_it_mult=2.0)
        assert rc == 0
        exp.append((t, P.vec(O.V_U).reshape(-1, 2)[ids].copy()))
    _check_rows(rows, exp, 2)
    assert np.abs(exp[-1][1][:, 0]).max() > 0  # the trace really has an x component that varies along the flap


@pytest.mark.parametrize("name", ["fsi3_neo_2d_explicit", "fsi3_neo_2d_implicit", "fsi3_linear_2d_shipped"])
def test_precice_v3_code_path_links_and_runs(tmp_path, name):
    """the host's -DMI_WITH_PRECICE branch (it includes <precice/precice.hpp> and talks to precice::Participant with the
    v3 signatures) LINKED against a test double of libprecice (tests/fake_precice: string_view / span arguments
    forwarded to the replay participant) and RUN: explicit and implicit (checkpointed) coupling and the linear model
    give bit-identical logs with the default build"""
    fake = os.path.join(ROOT, "tests", "fake_precice")
    subprocess.check_call(["make", "-C", fake])
    out = subprocess.run(["ldd", os.path.join(fake, "elasticity_precice")], capture_output=True, text=True).stdout
    assert "libprecice.so" in out and "libmi_elasticity.so" in out
    logs = []
    for exe in (os.path.join(HOST, "elasticity"), os.path.join(fake, "elasticity_precice")):
        d = tmp_path / os.path.basename(exe)
        d.mkdir()
        for f in ("parameters.prm", "precice-config.xml"):
            (d / f).write_text(open(os.path.join(CASES, name, f)).read())
        r = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        logs.append(open(d / "displacement.log").read())
    assert logs[0] == logs[1] and logs[0].count("\n") >= 3


@pytest.mark.parametrize("name", ["fsi3_neo_3d_q3", "fsi3_neo_3d_implicit"])
def test_multi_rank_coupling_goes_through_rank_zero(tmp_path, name):
    """BASELINE configuration 5 on several GPUs with a REAL participant [REF adapter.h:213-225, 346-385, 447-489]: under
    -DMI_WITH_PRECICE only rank 0 constructs precice::Participant(name, config, 0, 1); what it reads -- coupling data,
    isCouplingOngoing, the time-window size, the checkpoint requests of an implicit scheme -- reaches the other ranks through
    mi_comm_broadcast (Adapter::RankZeroParticipant), and only rank 0 writes.  Run against the libprecice test double:
      * one process, undecomposed (the reference's situation);
      * MI_SLABS=4: one process, four emulated slabs (the flap is cut along x);
      * four RANK THREADS through the library's RCCL branch against the RCCL test double (tests/fake_rccl/
        elasticity_ranks.cc -- a single-GPU box cannot host two real RCCL ranks).
    The test double counts its participants: exactly ONE may exist in every run.  Displacement logs agree to the linear
    tolerance (the decomposition changes the order of the sums), slabs and rank threads among themselves to 1e-13."""
    fake, frccl = os.path.join(ROOT, "tests", "fake_precice"), os.path.join(ROOT, "tests", "fake_rccl")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dealii-adapter_amd"), "-j4", "all"])
    subprocess.check_call(["make", "-C", fake])
    subprocess.check_call(["make", "-C", frccl, "libmi_elasticity_fakerccl.so", "ranks"])
    runs = {"one": ([os.path.join(fake, "elasticity_precice3d")], {}),
            "slabs": ([os.path.join(fake, "elasticity_precice3d")], {"MI_SLABS": "4"}),
            "ranks": ([os.path.join(frccl, "elasticity_ranks3d_precice"), "4"], {}),
            "ranks_replay": ([os.path.join(frccl, "elasticity_ranks3d"), "4"], {})}  # default build, same harness
    rows, outs = {}, {}
    for key, (cmd, env) in runs.items():
        d = tmp_path / key
        d.mkdir()
        for f in ("parameters.prm", "precice-config.xml"):
            (d / f).write_text(open(os.path.join(CASES, name, f)).read())
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, MI_FAKE_PRECICE_COUNT=str(d / "participants.txt"), **env))
        assert r.returncode == 0, key + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        rows[key] = [np.array(l.split(), dtype=float) for l in open(d / "displacement.log") if not l.startswith("#")]
        outs[key] = r.stdout
        if key != "ranks_replay":
            assert open(d / "participants.txt").read().split() == ["1"], key  # ONE participant, whatever the rank count
    n = len(rows["one"])
    assert n >= 2 and all(len(v) == n for v in rows.values())
    for key in ("slabs", "ranks", "ranks_replay"):
        for a, b in zip(rows["one"], rows[key]):
            assert a[0] == b[0] and np.abs(a[1:] - b[1:]).max() <= 1e-8 * np.abs(a[1:]).max(), key
    for a, b, c in zip(rows["slabs"], rows["ranks"], rows["ranks_replay"]):
        assert np.abs(a[1:] - b[1:]).max() <= 1e-13 * np.abs(a[1:]).max() and np.array_equal(b, c)
    assert "rank 0 couples" in outs["ranks"] and outs["ranks"].count("Number of coupling nodes") == 1  # ranks > 0 keep quiet
    assert (tmp_path / "ranks" / "out" / "solution-000.vtk").exists()


def test_rank_threads_stop_together_when_the_coupling_library_fails(tmp_path):
    """an error of the coupling library on the ONE preCICE-facing rank [REF adapter.h:213-225, 324-341] must end every rank:
    rank 0 catches it and the next collective call (mi_comm_broadcast behind Adapter::RankZeroParticipant) carries a status
    word, so the other ranks leave with an exception of their own instead of waiting in the all-reduce for ever.  Four rank
    threads against the RCCL test double; the replayed force trace misses vertices, which the participant reports from
    setMeshVertices on rank 0.  The harness must come back (its barrier would time out otherwise) with exit code 1, the
    original message from rank 0 and one 'stops with it' per other rank."""
    frccl = os.path.join(ROOT, "tests", "fake_rccl")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dealii-adapter_amd"), "-j4", "all"])
    subprocess.check_call(["make", "-C", frccl, "libmi_elasticity_fakerccl.so", "ranks"])
    name = "fsi3_neo_3d_q3"
    (tmp_path / "parameters.prm").write_text(open(os.path.join(CASES, name, "parameters.prm")).read())
    (tmp_path / "forces.txt").write_text("# t vertex fx fy fz\n0.01 0 0 -40 0\n0.01 1 0 -40 0\n")  # two of many vertices
    (tmp_path / "precice-config.xml").write_text(
        open(os.path.join(CASES, name, "precice-config.xml")).read().replace(
            "read-data = constant 0 -40 0", "read-data = vertex-trace forces.txt"))
    r = subprocess.run([os.path.join(frccl, "elasticity_ranks3d"), "4"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stderr.count("Exception on processing") == 4
    assert r.stderr.count("every vertex exactly once") == 1 and r.stderr.count("stops with it") == 3


def test_executable_nonlinear_implicit_checkpointing(tmp_path):
    """implicit coupling: 3 coupling iterations per window with save/reload of the 6 state vectors on the device"""
    name = "fsi3_neo_2d_implicit"
    _, rows = _run_case(name, "elasticity", tmp_path)
    get = _prm(name)
    P = O.Problem(_scenario_desc(get, 2))
    ids = P.interface_nodes
    dt, exp = float(get("Time step size")), []
    state_ids = (O.V_U, O.V_U_OLD, O.V_V, O.V_V_OLD, O.V_A, O.V_A_OLD)
    for w in range(2):
        saved = [P.vec(k).copy() for k in state_ids]  # requiresWritingCheckpoint at the window start
        for it in range(3):
            scale = 1.0 - 0.5 ** (it + 1) if it < 2 else 1.0
            P.set_interface_traction((0.0, -30.0 * scale))
            rc, _ = P.newmark_step(O.SOLVER_CG_SSOR, tol_lin=1e-12, max_it_mult=2.0)
            assert rc == 0
            if it < 2:  # requiresReadingCheckpoint
                for k, v in zip(state_ids, saved):
                    P.vec(k)[:] = v
        exp.append(((w + 1) * dt, P.vec(O.V_U).reshape(-1, 2)[ids].copy()))
    _check_rows(rows, exp, 2)


@pytest.mark.parametrize("name,windows,traction", [("pf_neo_3d_direct", 2, lambda k: (25.0, 0.0, 0.0)),
                                                   ("block_neo_3d_q2", 2, lambda k: (0.0, -2e3 * (k + 1) / 10.0, 0.0)),
                                                   ("fsi3_neo_3d_q3", 2, lambda k: (0.0, -40.0, 0.0))])
def test_executable_nonlinear_3d(tmp_path, name, windows, traction):
    _, rows = _run_case(name, "elasticity3d", tmp_path)
    get = _prm(name)
    P = O.Problem(_scenario_desc(get, 3))
    ids = P.interface_nodes
    dt, exp = float(get("Time step size")), []
    for k in range(windows):
        P.set_interface_traction(traction(k))
        rc, _ = P.newmark_step(O.SOLVER_DIRECT if get("Solver type") == "Direct" and P.n < 3000 else O.SOLVER_CG_SSOR,
                               tol_lin=1e-12, max_it_mult=2.0)
        assert rc == 0
        exp.append(((k + 1) * dt, P.vec(O.V_U).reshape(-1, 3)[ids].copy()))
    _check_rows(rows, exp, 3)
    if name == "block_neo_3d_q2":
        _check_vtk(tmp_path / "out" / "solution-000.vtk", P, zero=True)
        _check_vtk(tmp_path / "out" / "solution-001.vtk", P, zero=False)  # timestep 2 / Output interval 2
        # the switch back to linear sub-cells (readers older than ParaView 5.5, :1221): p^3 = 8 hexahedra (12) per patch,
        # same points and fields
        sub = tmp_path / "linear"
        sub.mkdir()
        _run_case(name, "elasticity3d", sub, env={"MI_VTK_LINEAR_CELLS": "1"})
        a = open(tmp_path / "out" / "solution-001.vtk").read().split("\n")
        b = open(sub / "out" / "solution-001.vtk").read().split("\n")
        kb = next(i for i, l in enumerate(b) if l.startswith("CELL_TYPES"))
        assert int(b[kb].split()[1]) == 8 * P.ncells and set(b[kb + 1:kb + 1 + 8 * P.ncells]) == {"12"}
        ia, ib = (next(i for i, l in enumerate(t) if l.startswith("CELLS")) for t in (a, b))
        assert a[:ia] == b[:ib]  # header and points
        ja, jb = (next(i for i, l in enumerate(t) if l.startswith("POINT_DATA")) for t in (a, b))
        assert a[ja:] == b[jb:]  # fields
        # the executable's switch for the matrix-free fine level (tuning "fine_level" 1 + "mf_diag_lag" 1): same windows
        mf = tmp_path / "mf"
        mf.mkdir()
        _, rows_mf = _run_case(name, "elasticity3d", mf, env={"MI_FINE_LEVEL": "1"})
        _check_rows(rows_mf, exp, 3)


def test_executable_on_emulated_slabs(tmp_path):
    """the decomposition reaches the executables through the environment (mi/device_vector.h): MI_SLABS=2 cuts the block
    into two z-slabs inside one process; same banner plus the decomposition line, same displacement log and VTK output as
    the undecomposed run to the linear tolerance.  (One process per GPU over RCCL -- tools/launch_elasticity.py, MI_RANK /
    MI_WORLD_SIZE / MI_UID_FILE -- takes the same path in mi::Device and cannot run on a single-GPU box.)"""
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    out1, rows1 = _run_case("block_neo_3d_q2", "elasticity3d", tmp_path / "one")
    out2, rows2 = _run_case("block_neo_3d_q2", "elasticity3d", tmp_path / "two", env={"MI_SLABS": "2"})
    assert "2 slabs emulated on one GPU" in out2 and "slabs emulated" not in out1
    assert len(rows1) == len(rows2) > 0
    for a, b in zip(rows1, rows2):
        assert a[0] == b[0] and np.abs(a[1:] - b[1:]).max() <= 1e-9 * np.abs(a[1:]).max()
    v1 = open(tmp_path / "one" / "out" / "solution-001.vtk").read().split()
    v2 = open(tmp_path / "two" / "out" / "solution-001.vtk").read().split()
    assert len(v1) == len(v2)


def test_executable_cuts_the_flap_along_x(tmp_path):
    """BASELINE configuration 5's geometry in 3D: the FSI3 flap has 18 x 3 x 1 cells (nonlinear_elasticity.cc:189-205), so
    until round 3 it could not be decomposed at all (slabs were cut along z only).  Four parts along x, chosen by the
    library itself, against the undecomposed run: same displacement log to the linear tolerance."""
    (tmp_path / "one").mkdir()
    (tmp_path / "four").mkdir()
    out1, rows1 = _run_case("fsi3_neo_3d_q3", "elasticity3d", tmp_path / "one")
    out4, rows4 = _run_case("fsi3_neo_3d_q3", "elasticity3d", tmp_path / "four", env={"MI_SLABS": "4"})
    assert "4 slabs emulated on one GPU" in out4
    assert len(rows1) == len(rows4) > 0
    for a, b in zip(rows1, rows4):
        assert a[0] == b[0] and np.abs(a[1:] - b[1:]).max() <= 1e-8 * np.abs(a[1:]).max()


def test_launcher_rank_environment_is_validated(tmp_path):
    """MI_WORLD_SIZE without the id file: the executable refuses loudly instead of running undecomposed"""
    for f in ("parameters.prm", "precice-config.xml"):
        (tmp_path / f).write_text(open(os.path.join(CASES, "block_neo_3d_q2", f)).read())
    out = subprocess.run([os.path.join(HOST, "elasticity3d")], cwd=tmp_path, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, MI_WORLD_SIZE="2", MI_RANK="0"))
    assert out.returncode == 1 and "MI_UID_FILE" in out.stderr


def test_launcher_ends_all_ranks_when_one_fails(tmp_path):
    """tools/launch_elasticity.py -n 2 on a box with ONE GPU: rank 1 has no device and exits with the executable's
    message; the launcher must end rank 0 (which waits for its partner in RCCL) instead of hanging, and report failure"""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a single-GPU box")
    for f in ("parameters.prm", "precice-config.xml"):
        (tmp_path / f).write_text(open(os.path.join(CASES, "block_neo_3d_q2", f)).read())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_elasticity.py"), "-n", "2", "parameters.prm"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=100)
    assert out.returncode == 1 and "out of range" in out.stderr


def _check_vtk_lagrange_cells(txt, ref, ncells, npc):
    """round 6 (f-3's last gap): ONE higher-order cell per patch, as `flags.write_higher_order_cells = true` makes DataOut
    write them [REF nonlinear_elasticity.cc:1222-1225]: type 72 = VTK_LAGRANGE_HEXAHEDRON, (p+1)^3 = 27 point ids each -- a
    permutation of the patch's own points -- in VTK's Lagrange order, checked GEOMETRICALLY on the reference coordinates of
    the listed points: 8 corners in the linear hexahedron's order, 12 edge midpoints (bottom face x y x y, top face, the
    vertical edges (x0,y0) (x1,y0) (x0,y1) (x1,y1)), 6 face centres (x- x+ y- y+ z- z+), the body centre"""
    i = next(k for k, l in enumerate(txt) if l.startswith("CELLS"))
    n, size = (int(v) for v in txt[i].split()[1:3])
    assert n == ncells and size == ncells * (npc + 1)
    cells = np.array([l.split() for l in txt[i + 1:i + 1 + n]], dtype=np.int64)
    assert cells.shape == (ncells, npc + 1) and np.all(cells[:, 0] == npc)
    j = next(k for k, l in enumerate(txt) if l.startswith("CELL_TYPES"))
    assert int(txt[j].split()[1]) == ncells and all(int(t) == 72 for t in txt[j + 1:j + 1 + ncells])
    # unit-cell position (0, 1/2, 1 per direction) of every VTK slot of the quadratic Lagrange hexahedron
    c8 = [(0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2)]
    e12 = [(1, 0, 0), (2, 1, 0), (1, 2, 0), (0, 1, 0), (1, 0, 2), (2, 1, 2), (1, 2, 2), (0, 1, 2),
           (0, 0, 1), (2, 0, 1), (0, 2, 1), (2, 2, 1)]
    f6 = [(0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2)]
    want = np.array(c8 + e12 + f6 + [(1, 1, 1)])
    for c in range(ncells):
        ids = cells[c, 1:]
        assert sorted(ids) == list(range(c * npc, (c + 1) * npc))  # a permutation of the patch
        X = ref[ids]
        lo, hi = X.min(0), X.max(0)
        assert np.array_equal(np.rint(2 * (X - lo) / (hi - lo)).astype(int), want), c


def _check_vtk(path, P, zero):
    """patch-wise VTK with displacement + strain_ij scalars (postprocessor.h:46-111)"""
    txt = open(path).read().split("\n")
    npc, ncells = 27, P.ncells
    i = next(k for k, l in enumerate(txt) if l.startswith("POINTS"))
    npts = int(txt[i].split()[1])
    assert npts == ncells * npc
    pts = np.array([l.split() for l in txt[i + 1:i + 1 + npts]], dtype=float)
    j = next(k for k, l in enumerate(txt) if l.startswith("VECTORS displacement"))
    disp = np.array([l.split() for l in txt[j + 1:j + 1 + npts]], dtype=float)
    fields = {}
    for k, l in enumerate(txt):
        if l.startswith("SCALARS strain_"):
            fields[l.split()[1]] = np.array(txt[k + 2:k + 2 + npts], dtype=float)
    assert sorted(fields) == sorted("strain_" + a + b for a in "xyz" for b in "xyz")
    _check_vtk_lagrange_cells(txt, pts - disp, ncells, npc)
    if zero:
        assert np.all(disp == 0) and all(np.all(f == 0) for f in fields.values())
        return
    # points are the displaced support points; displacement matches the oracle node by reference coordinate
    ref = pts - disp
    X, U = P.coords, P.vec(O.V_U).reshape(-1, 3)
    key = {tuple(np.round(x, 7) + 0.0): k for k, x in enumerate(X)}
    idx = np.array([key[tuple(np.round(x, 7) + 0.0)] for x in ref])
    assert np.abs(disp - U[idx]).max() / np.abs(U).max() < 1e-7  # 12 significant digits in the ASCII file
    for a in "xyz":
        for b in "xyz":
            assert np.array_equal(fields["strain_" + a + b], fields["strain_" + b + a])
    # strain VALUES: sym(grad u) with the gradient taken in the output mapping (MappingQEulerian: the displaced Q2
    # geometry, nonlinear_elasticity.cc:1232-1236, postprocessor.h:60-75), recomputed per cell patch from the ORACLE's
    # displacement with the tensor-product Lagrange basis on the patch's own 27 support points
    def lag(nodes, x):
        N, dN = np.ones(3), np.zeros(3)
        for a in range(3):
            for m in range(3):
                if m != a:
                    N[a] *= (x - nodes[m]) / (nodes[a] - nodes[m])
            for k in range(3):
                if k != a:
                    t = 1.0 / (nodes[a] - nodes[k])
                    for m in range(3):
                        if m not in (a, k):
                            t *= (x - nodes[m]) / (nodes[a] - nodes[m])
                    dN[a] += t
        return N, dN

    unit = np.array([0.0, 0.5, 1.0])
    tab = [lag(unit, x) for x in unit]
    worst = 0.0
    for c in range(0, npts, npc):
        blk = slice(c, c + npc)
        Xr, nodes_c = ref[blk], idx[c:c + npc]
        lo, hi = Xr.min(0), Xr.max(0)
        xi = np.rint(2 * (Xr - lo) / (hi - lo)).astype(int)  # 0, 1, 2 per direction
        Uc = U[nodes_c]
        for q in range(npc):
            dN = np.zeros((npc, 3))
            for a in range(npc):
                for k in range(3):
                    v = 1.0
                    for d in range(3):
                        v *= tab[xi[q, d]][1 if d == k else 0][xi[a, d]]
                    dN[a, k] = v
            H = Uc.T @ dN                 # du_i / dxi_j
            Jx = (Xr + Uc).T @ dN         # dx_i / dxi_j of the displaced Q2 geometry
            E = H @ np.linalg.inv(Jx)
            E = 0.5 * (E + E.T)
            got = np.array([[fields["strain_" + a + b][c + q] for b in "xyz"] for a in "xyz"])
            worst = max(worst, np.abs(got - E).max())
        if c >= 40 * npc:  # 40 patches are plenty (python loops)
            break
    scale = max(np.abs(fields["strain_" + a + b]).max() for a in "xyz" for b in "xyz")
    assert 0 < scale < 0.1 and worst / scale < 1e-6, (worst, scale)


def _linear_pair(desc):
    P = O.LinearProblem(desc)
    G = M.Context(dim=desc.dim, degree=desc.degree, reps=tuple(desc.reps)[:desc.dim], lo=tuple(desc.lo)[:desc.dim],
                  hi=tuple(desc.hi)[:desc.dim], face_role=list(desc.face_role), mu=desc.mu, nu=desc.nu, rho=desc.rho,
                  body_force=tuple(desc.body_force), delta_t=desc.delta_t)
    L = M.lib()
    L.mi_linear_setup.argtypes = [C.c_void_p, C.c_double]
    L.mi_linear_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mi_linear_matrix_get_csr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double)]
    assert L.mi_linear_setup(G.h, desc.theta) == 0, L.mi_last_error(G.h)
    return P, G


def _linear_matrix(G, which):
    import scipy.sparse as sp
    rp = np.zeros(G.n + 1, dtype=np.int64)
    col = np.zeros(G.nnz, dtype=np.int32)
    val = np.zeros(G.nnz)
    assert M.lib().mi_linear_matrix_get_csr(G.h, which, rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                            col.ctypes.data_as(C.POINTER(C.c_int32)), M._dp(val)) == 0
    return sp.csr_matrix((val, col, rp), shape=(G.n, G.n))


@pytest.mark.parametrize("dim,p,reps", [(2, 3, (6, 2)), (3, 1, (5, 3, 2)), (3, 2, (2, 2, 2))])
def test_linear_model_matrices_and_steps(dim, p, reps):
    """K, M, stepping matrix against the oracle (1e-12), then 5 theta-steps incl. consistent loading, body force,
    warm-started CG; both the 'Stress' and the 'Force' read-data paths"""
    roles = [O.FACE_CLAMPED, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_INTERFACE, O.FACE_ZCLAMP, O.FACE_ZCLAMP]
    desc = O.make_desc(dim=dim, degree=p, reps=reps, hi=tuple(0.2 * r for r in reps), face_role=roles, mu=0.5e6, nu=0.4,
                       rho=1000.0, body_force=(0.0, -9.81, 0.0), delta_t=0.005, theta=0.6)
    P, G = _linear_pair(desc)
    for which in (0, 1):
        A_o, A_g = P.matrix(which), _linear_matrix(G, which)
        assert np.array_equal(A_o.indices, A_g.indices)
        assert np.abs(A_g.data - A_o.data).max() / np.abs(A_o.data).max() < 1e-12
    ids = P.interface_nodes
    rng = np.random.default_rng(5)
    for step in range(5):
        consistent = step != 3
        t = 100.0 * rng.standard_normal((len(ids), dim))
        P.vec(O.L_STRESS)[:] = 0
        for c in range(dim):
            P.vec(O.L_STRESS)[ids * dim + c] = t[:, c]
        G.set_interface_traction(t)
        rc, its_o, res_o = P.step(O.SOLVER_DIRECT, consistent)
        assert rc == 0
        its, res = C.c_int(0), C.c_double(0)
        rc = M.lib().mi_linear_step(G.h, int(consistent), 1e-12, G.n * 4, C.byref(its), C.byref(res))
        assert rc == 0, M.lib().mi_last_error(G.h)
        assert res.value <= 1e-12 and its.value > 0
        if step == 0:
            A_o, A_g = P.matrix(3), _linear_matrix(G, 2)  # system matrix with boundary values applied
            assert np.abs(A_g.data - A_o.data).max() / np.abs(A_o.data).max() < 1e-12
        for vo, vg in ((O.L_D, 0), (O.L_V, 2), (O.L_V_OLD, 3), (O.L_STRESS_OLD, 4)):
            ref = P.vec(vo)
            assert np.abs(G.get(vg) - ref).max() / max(np.abs(ref).max(), 1e-300) < TOL
    assert np.all(G.get(2)[P.constrained] == 0)


@pytest.mark.parametrize("name,exe,dim", [("fsi3_linear_2d_shipped", "elasticity", 2),
                                          ("pf_linear_2d", "elasticity", 2),
                                          ("fsi3_linear_3d_shipped", "elasticity3d", 3),
                                          ("block_linear_3d_q1_cg", "elasticity3d", 3)])
def test_executable_linear(tmp_path, name, exe, dim):
    """config 1 (shipped settings: linear, Direct, degree 3, FSI3) under -DDIM=2 and -DDIM=3 (parameters.prm:19-22,
    CMakeLists.txt:15-18), and a small config 2 (3D Q1 block, CG)"""
    stdout, rows = _run_case(name, exe, tmp_path)
    get = _prm(name)
    P = O.LinearProblem(_scenario_desc(get, dim, theta=0.5))
    ids = P.interface_nodes
    oracle_direct = P.n < 4000  # the oracle's banded LU; beyond that its CG + SSOR at the reference's absolute 1e-10
    dt, exp = float(get("Time step size")), []
    for k in range(len(rows)):
        t = ((0.0, -40.0 * min(1.0, (k + 1) / 2.0), 0.0)[:dim] if "shipped" in name else
             (30.0 * min(1.0, (k + 1) / 2.0), 0.0) if name == "pf_linear_2d" else (0.0, -200.0, 0.0))
        P.vec(O.L_STRESS)[:] = 0
        for c in range(dim):
            P.vec(O.L_STRESS)[ids * dim + c] = t[c]
        rc, _, _ = P.step(O.SOLVER_DIRECT if oracle_direct else O.SOLVER_CG_SSOR, True)
        assert rc == 0
        exp.append(((k + 1) * dt, P.vec(O.L_D).reshape(-1, dim)[ids].copy()))
    assert len(rows) >= 3
    # CG with the reference's absolute tolerance 1e-10 on both sides: velocities agree to ~1e-10/|A|, so compare at 1e-6
    for r, (t, u) in zip(rows, exp):
        got = r[1:].reshape(-1, dim)
        assert np.abs(got - u).max() / np.abs(u).max() < (1e-8 if get("Solver type") == "Direct" and oracle_direct else 1e-6)
    assert "No of iterations" in stdout and "Solve system" in stdout
