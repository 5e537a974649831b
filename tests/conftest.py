import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def ns():
    from source_amd import api
    return api


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def m1m(ns):
    """The bench's `flat` workload: ONE mesh of 1 048 576 triangles (scenes.build_flat; host build ~4 s, shared by the session)."""
    from source_amd import scenes
    v, t = scenes.displaced_sphere(512)
    return ns.Mesh(v, t, smoothing=False, closed=True), v, t


@pytest.fixture(scope="session")
def m70k(ns):
    """The ~70k-triangle stand-in mesh (host build only; no GPU involved)."""
    from source_amd import scenes
    v, t = scenes.displaced_sphere(132)
    return ns.Mesh(v, t, smoothing=False), v, t
