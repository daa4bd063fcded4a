import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
ROBOTS = ["iiwa7", "atlas30", "mixed5", "quad12"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=ROBOTS)
def robot_name(request):
    return request.param


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def robots():
    from gridcodegenerator_amd.robots import get_robot
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = get_robot(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def tables(robots):
    from oracle import rbd_oracle as O
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = O.RobotTables(robots(name))
        return cache[name]
    return get


def make_inputs(n, K, seed):
    """SURVEY.md section 8(d) input distribution; fp32-representable values."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (K, n)).astype(np.float32)
    qd = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    u = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    return q, qd, u


def relerr(got, ref):
    """(norm-wise relative error, worst element-wise relative error over non-tiny reference entries)."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-300)
    err = np.abs(got - ref)
    mask = np.abs(ref) > 1e-6 * scale
    worst = (err[mask] / np.abs(ref[mask])).max() if mask.any() else 0.0
    return err.max() / scale, worst


def elementwise_err(got, ref, floor=1e-3):
    """Worst ELEMENT-WISE relative error over the entries whose reference is at least `floor` of the batch scale max|ref|: the
    entries a consumer can tell from zero.  (Entries near a zero crossing carry the absolute round-off of the scale; their relative
    error is unbounded by construction and is what relerr()[1] reports.)"""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-300)
    mask = np.abs(ref) >= floor * scale
    return float((np.abs(got - ref)[mask] / np.abs(ref[mask])).max()) if mask.any() else 0.0
