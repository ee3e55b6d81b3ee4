import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def snappy_raw(oracle):
    """Raw Snappy corpus = oracle-decoded fixtures, each verified against its .hash."""
    import glob
    import hashlib
    out = {}
    for f in sorted(glob.glob(os.path.join(GOLDEN, "snappy", "*.lzfse"))):
        name = os.path.basename(f)[:-6]
        raw = oracle.decode(open(f, "rb").read())
        assert hashlib.sha256(raw).digest() == open(f[:-6] + ".hash", "rb").read()
        out[name] = raw
    return out
