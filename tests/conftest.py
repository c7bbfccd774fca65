import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's small ops are slower with one OpenMP thread per core of a 256-core host than with 32
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def golden_sd(z, prefix="sd."):
    import torch
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
