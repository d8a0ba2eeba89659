import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell_emissive():
    from zetaray_amd import scene_io
    return scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))


@pytest.fixture(scope="session")
def oracle_emissive(cornell_emissive):
    from oracle import zro
    return zro.OracleScene(cornell_emissive)
