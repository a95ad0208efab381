"""CPU tests of the host-side boundary code: the C++ unit-test binary (parameter reader, Adapter::Time,
replay participant, Adapter call order) and the executables' command-line contract without a device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-adapter_amd", "host")


@pytest.fixture(scope="module")
def host_built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dealii-adapter_amd"), "-j4", "all"])
    subprocess.check_call(["make", "-C", HOST, "-j4", "all"])


def test_host_unit_tests(host_built, tmp_path):
    out = subprocess.run([os.path.join(HOST, "test_host")], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "HOST TESTS OK" in out.stdout


def test_missing_parameter_file_exits_with_1(host_built, tmp_path):
    """exception -> 'Exception on processing:' block on stderr and exit code 1 (elasticity.cc:101-126)"""
    out = subprocess.run([os.path.join(HOST, "elasticity"), "nope.prm"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 1
    assert "Exception on processing:" in out.stderr and "Aborting!" in out.stderr
    assert "running with 1 thread" in out.stdout  # banner comes first (:32-44)


def test_force_data_is_rejected_for_neo_hookean(host_built, tmp_path):
    """nonlinear_elasticity.cc:83-87; fails before any device work"""
    case = os.path.join(ROOT, "tests", "cases", "fsi3_neo_2d_explicit")
    prm = open(os.path.join(case, "parameters.prm")).read().replace("= Stress", "= Force")
    (tmp_path / "parameters.prm").write_text(prm)
    (tmp_path / "precice-config.xml").write_text(open(os.path.join(case, "precice-config.xml")).read())
    out = subprocess.run([os.path.join(HOST, "elasticity")], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 1 and "doesn't support 'Force' data reading" in out.stderr
    assert os.path.isdir(tmp_path / "out")  # the output folder is created before the solver starts (:56-81)


def test_no_device_is_a_loud_error(host_built, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    case = os.path.join(ROOT, "tests", "cases", "fsi3_neo_2d_explicit")
    for f in ("parameters.prm", "precice-config.xml"):
        (tmp_path / f).write_text(open(os.path.join(case, f)).read())
    out = subprocess.run([os.path.join(HOST, "elasticity")], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr


def test_host_compiles_against_precice_v3_api():
    """-DMI_WITH_PRECICE swaps the replay participant for <precice/precice.hpp>.  libprecice is not installed here, so
    the sources are type-checked (g++ -fsyntax-only) against a declaration-only copy of the v3 API restricted to the
    14 calls the reference makes (tests/precice_api/precice/precice.hpp): string_view / span<const double> /
    span<VertexID> arguments, const-qualification and return types of every call site"""
    import subprocess
    host = os.path.join(ROOT, "dealii-adapter_amd", "host")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    for dim in (2, 3):
        for src in ("elasticity.cc", "source/nonlinear_elasticity/nonlinear_elasticity.cc",
                    "source/linear_elasticity/linear_elasticity.cc"):
            cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-parameter", "-DDIM=%d" % dim,
                   "-DMI_WITH_PRECICE", "-I" + os.path.join(ROOT, "tests", "precice_api"), "-Iinclude",
                   "-I" + os.path.join(ROOT, "include"), "-I.", "-I" + os.path.join(rocm, "include"),
                   "-D__HIP_PLATFORM_AMD__", src]
            out = subprocess.run(cmd, cwd=host, capture_output=True, text=True)
            assert out.returncode == 0 and "error" not in out.stderr, out.stderr[-3000:]


def test_precice_v3_build_links_against_the_test_double(host_built, tmp_path):
    """-DMI_WITH_PRECICE build linked against tests/fake_precice/libprecice.so (the v3 signatures forwarded to the replay
    participant); without a device it still gets as far as the reference does before the solver starts"""
    fake = os.path.join(ROOT, "tests", "fake_precice")
    subprocess.check_call(["make", "-C", fake])
    out = subprocess.run(["ldd", os.path.join(fake, "elasticity_precice")], capture_output=True, text=True).stdout
    assert "libprecice.so" in out
    r = subprocess.run([os.path.join(fake, "elasticity_precice"), "nope.prm"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Exception on processing:" in r.stderr
