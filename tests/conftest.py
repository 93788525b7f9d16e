import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is test infrastructure; build it on demand (gcc only, ~1 s).
    so = os.path.join(ROOT, "oracle", "libld_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libld_oracle.so"])
    _gpu_present()


_GPU = None


def _gpu_present():
    # Cached, and asked for before any test module is imported: torch ships its own HIP runtime, and
    # the runtime that libdeflate_hip.so links finds no device when torch's is initialised after it.
    global _GPU
    if _GPU is None:
        try:
            import torch

            _GPU = bool(torch.cuda.is_available())
        except Exception:
            _GPU = False
    return _GPU


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
