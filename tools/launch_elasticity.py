#!/usr/bin/env python3
"""Start the `elasticity` executable on N GPUs of one node: one process per GPU, the box cut into N slabs (along the direction with most cell layers), ghost planes
and reductions over RCCL (the reference is single-rank, adapter.h:152-154; this is the launcher DESIGN.md section 6
describes).  Every process runs the same program on global views of the interface; ONE process talks to the coupling
library: rank 0 owns the precice::Participant (replay participant or, in a -DMI_WITH_PRECICE build, libprecice) and the other
ranks receive what it reads through mi_comm_broadcast (host/include/adapter/rank_zero_participant.h); rank 0 prints and
writes the output files.

  python tools/launch_elasticity.py -n 8 [--exe dealii-adapter_amd/host/elasticity3d] [parameters.prm]

The processes find each other through three variables this script sets: MI_WORLD_SIZE, MI_RANK (MI_LOCAL_RANK = device)
and MI_UID_FILE, a scratch file in which rank 0 leaves the 128-byte RCCL id (host/include/mi/device_vector.h).
`MI_SLABS=N elasticity …` (no launcher) runs the same decomposition inside ONE process on one GPU."""
import argparse
import os
import subprocess
import sys
import tempfile
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--ranks", type=int, required=True)
    ap.add_argument("--exe", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dealii-adapter_amd", "host",
                                                  "elasticity3d"))
    ap.add_argument("prm", nargs="?", default="parameters.prm")
    a = ap.parse_args()
    if a.ranks < 1:
        ap.error("need at least one rank")
    with tempfile.TemporaryDirectory(prefix="mi_uid_") as d:
        uid = os.path.join(d, "rccl_unique_id")
        procs = []
        for r in range(a.ranks):
            env = dict(os.environ, MI_WORLD_SIZE=str(a.ranks), MI_RANK=str(r), MI_LOCAL_RANK=str(r), MI_UID_FILE=uid,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([a.exe, a.prm], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
        rc = 0
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):  # one rank failed: the others may wait in a collective for ever
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.05)
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
