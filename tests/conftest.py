import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `pytest tests` without -m on a CPU-only box: skip the gpu tests instead of failing
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def host_is_intel():
    try:
        return "GenuineIntel" in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLD, "frames.json")) as f:
        return json.load(f)["cases"]


def load_spheres(key):
    from oracle import SPHERE_DTYPE
    if key == "t0":
        return np.load(os.path.join(GOLD, "spheres_t0.npy"))
    if key == "none":
        return np.zeros(0, SPHERE_DTYPE)
    return np.load(os.path.join(GOLD, "levels", key + "_spheres.npy"))


def level_path(name):
    return os.path.join(GOLD, "levels", name + ".txt")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
