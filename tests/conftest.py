import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import mcrt_pkg

    return mcrt_pkg.load()


@pytest.fixture(scope="session")
def po():
    import pyoracle

    return pyoracle


@pytest.fixture(scope="session")
def oracle(po):
    return po.Oracle()


@pytest.fixture(scope="session")
def ref(po):
    """The reference's own code, where oracle/_ref has been built (never required)."""
    if not po.Ref.available():
        pytest.skip("oracle/_ref/libmcrt_ref.so not built (needs /root/reference)")
    return po.Ref()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def scene_of(pkg):
    def _load(name):
        return pkg.FlatScene.load(os.path.join(GOLDEN, f"scene_{name}.npz"))

    return _load


@pytest.fixture(scope="session")
def backend(pkg):
    """One GPU context for the whole session (GPU tests only)."""
    b = pkg.Backend(0)
    yield b
    b.close()


SCENES = ["cornell", "config2", "config3", "config4_2k", "hexagon", "cornell_point", "cornell_sphere_light", "three_lights"]
