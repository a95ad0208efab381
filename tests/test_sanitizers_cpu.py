"""CPU sanitizer run ("race detection / sanitizers"): AddressSanitizer + UndefinedBehaviorSanitizer
builds of the host-side code -- the boundary classes' unit tests (parameter reader, Adapter, RankZeroParticipant with rank
threads, replay participant), the host-only entry points of the C-ABI (mi_partition_*: mi::SlabPartition, mi::HostMesh)
and the oracle -- each run through a driver under tests/asan.  GPU sanitizers are not available on the MI355X pool."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "asan"), "run"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for line in ("HOST TESTS OK", "PARTITION SANITIZER RUN OK", "ORACLE SANITIZER RUN OK"):
        assert line in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
