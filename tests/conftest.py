import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """a fresh checkout has no binaries: build the HIP library, the host executables and the oracle once (hipcc
    cross-compiles without a GPU).  This builds the product, it does not replace it: there is no CPU fallback."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "dealii-adapter_amd")
    need = [os.path.join(pkg, "libmi_elasticity.so"), os.path.join(pkg, "host", "elasticity"),
            os.path.join(pkg, "host", "elasticity3d"), os.path.join(pkg, "host", "test_host"),
            os.path.join(root, "oracle", "liboracle.so")]
    if all(os.path.exists(f) for f in need):
        return
    for d, tgt in ((pkg, "all"), (os.path.join(pkg, "host"), "all"), (os.path.join(root, "oracle"), "liboracle.so")):
        subprocess.check_call(["make", "-C", d, "-j4", tgt])


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. in the build container
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_pkg():
    """import dealii-adapter_amd/ (the directory name is not a valid module name) as dealii_adapter_amd"""
    import importlib.util
    name = "dealii_adapter_amd"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "dealii-adapter_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
