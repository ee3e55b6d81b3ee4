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


@pytest.fixture(scope="session")
def diag_ctx():
    """Context on the diagnostic build of the library (liblzfse_mi_diag.so): the LZFSE_MI_OPT_DIAG_* options that force a
    code path exist there only. Tests that do not force anything use the product library."""
    import lzfse_rust_amd as m
    return m.Context(0, diag=True)
