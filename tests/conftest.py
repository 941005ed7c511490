import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """The native libraries normally travel with the tree (built by __graft_entry__.build());
    if one is missing and the toolchain is present, build it once instead of failing every test."""
    need = [os.path.join(ROOT, "tristage-rag_amd", "libtristage.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import shutil
    import subprocess
    if shutil.which("make") is None:
        return
    if not os.path.exists(need[0]) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tristage-rag_amd", "csrc")], check=False)
    if not os.path.exists(need[1]) and shutil.which("gcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False)
