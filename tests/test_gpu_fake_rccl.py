"""The library's RCCL branch on a one-GPU box.  RCCL refuses two ranks on one device, so the ranks are threads of
one process and RCCL is replaced by a test double with the real prototypes (tests/fake_rccl/fake_rccl.cpp): what
runs is the product's own multi-rank code -- one slab per rank, `ncclSend/ncclRecv` halo groups on the communication
stream with its events, `ncclAllReduce` of scalars, vectors and global buffers, the order of collective calls on every
rank (a mismatch times out in the double instead of hanging) -- compared with the undecomposed and the emulated runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dealii-adapter_amd"), "-j4", "all"])
    subprocess.check_call(["make", "-C", FAKE])
    out = subprocess.run(["ldd", os.path.join(FAKE, "libmi_elasticity_fakerccl.so")], capture_output=True, text=True).stdout
    assert "rccl" not in out  # the real library is not in the picture
    return True


@pytest.mark.parametrize("world,dim,p,reps,overlap,ebe", [(2, 3, 2, "3,2,5", 1, 0), (3, 3, 2, "3,3,7", 1, 0),
                                                          (4, 3, 1, "4,3,9", 1, 0), (2, 3, 2, "3,2,5", 0, 0),
                                                          (3, 2, 3, "4,9", 1, 0), (8, 3, 2, "3,3,17", 1, 0),
                                                          (3, 3, 2, "3,3,7", 1, 1), (3, 3, 2, "3,3,7", 1, 2),
                                                          (2, 3, 2, "4,3,6", 0, 2),
                                                          # most cell layers along x / y: the lattice lies rotated over the box
                                                          (3, 3, 2, "7,2,3", 1, 0), (4, 3, 1, "3,9,2", 1, 0),
                                                          (3, 2, 2, "9,4", 1, 0), (4, 3, 2, "9,2,2", 0, 2)])
def test_rank_threads_through_the_rccl_branch(fake_lib, world, dim, p, reps, overlap, ebe):
    """ebe = 1 / 2: the multigrid smoother on the stored element tangents / matrix-free, as on big meshes"""
    _run_ranks(world, dim, p, reps, overlap, ebe, -1)


@pytest.mark.parametrize("world,dim,p,reps", [(3, 3, 2, "3,3,13"), (4, 3, 1, "4,3,17"), (3, 2, 2, "4,19"), (2, 3, 2, "9,3,3")])
def test_rank_threads_with_a_distributed_first_coarsened_level(fake_lib, world, dim, p, reps):
    """round 5: the first COARSENED multigrid level cut into slabs of its own (cuts induced by the finer level's; restriction
    and state through partial results on ghost planes, team_halo_accumulate) -- forced on these small meshes by
    tuning "mg_dist_nodes" 0; the same checks as above: every rank the same bits, the emulated slabs to 1e-9, one GPU to the
    linear tolerance, iteration counts within one"""
    _run_ranks(world, dim, p, reps, 1, 0, 0)


@pytest.mark.parametrize("world,reps,overlap", [(3, "3,3,7", 1), (2, "4,3,6", 0), (3, "7,2,3", 1), (8, "3,3,17", 1)])
def test_rank_threads_with_the_fine_level_matrix_free(fake_lib, world, reps, overlap):
    """round 6, tuning "fine_level" 1 through the RCCL branch: no assembled fine tangent on any rank -- the CG's product is
    mf_spmv in two launches around the ghost-plane send/recv (inner cell layers while the planes travel) + the gather with the
    partials of p.q, the tangent pass writes records / residual / diagonal blocks of the local cells (ghost layer included);
    every rank the same bits, the emulated slabs to 1e-9, one GPU to the linear tolerance, iteration counts within one"""
    _run_ranks(world, 3, 2, reps, overlap, 2, -1, fine=1)


def _run_ranks(world, dim, p, reps, overlap, ebe, dist_nodes, fine=0):
    out = subprocess.run([sys.executable, os.path.join(FAKE, "run_ranks.py"), str(world), str(dim), str(p), reps, str(overlap),
                          str(ebe), str(dist_nodes), str(fine)], capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.stdout[-2000:], out.stderr[-3000:])
    r = json.loads(lines[-1])
    assert r["ok"], r
    assert r["rank_spread"] == 0.0  # every rank returns the same global arrays, bit for bit
    assert r["broadcast_ok"]  # mi_comm_broadcast: rank 0's values on every rank, whatever the others passed in
    for k, v in r["vs_single"].items():
        assert v < 1e-7, (k, v, r)  # CG tolerance 1e-10 on both sides
    for k, v in r["vs_emulated"].items():
        assert v < 1e-9, (k, v, r)  # same slabs, same reduction order up to the all-reduce's summation
    # levels cut into slabs: the fine level (+ the Q1 level on the same cells for degree > 1), + the first coarsened level
    # where "mg_dist_nodes" 0 forces it
    base = 2 if p > 1 else 1
    assert r["mg_dist_levels"] == [base + (1 if dist_nodes == 0 else 0)] * 2 + [0], r["mg_dist_levels"]
    for its in r["its_ranks"]:
        assert its == r["its_ranks"][0]
        assert all(abs(a - b) <= 1 for a, b in zip(its, r["its_emulated"]))
