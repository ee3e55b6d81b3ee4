"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/*.h
declares, status codes agree with the oracle's, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from lzfse_rust_amd import build
    build.build()
    from lzfse_rust_amd import _native
    return _native.lib()


def _declared_functions(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lzfse_mi_[a-z_0-9]+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib):
    names = _declared_functions(os.path.join(ROOT, "include", "lzfse_mi.h"))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n


def test_status_codes_match_oracle():
    def enum(path, prefix):
        txt = open(path).read()
        return {k[len(prefix):]: int(v) for k, v in re.findall(r"\b(%s[A-Z_0-9]+)\s*=\s*(\d+)" % prefix, txt)}
    mi = enum(os.path.join(ROOT, "include", "lzfse_mi.h"), "LZFSE_MI_")
    lo = enum(os.path.join(ROOT, "oracle", "lzfse_oracle.h"), "LZO_")
    assert len(lo) > 25
    for k, v in lo.items():
        assert mi[k] == v, k


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.lzfse_mi_create(0, C.byref(h)) == 10  # LZFSE_MI_NO_DEVICE, never a CPU fallback
    import lzfse_rust_amd as m
    with pytest.raises(m.LzfseError):
        m.LzfseDecoder()


def test_encode_bound_and_decode_size(lib, oracle, snappy_raw):
    import lzfse_rust_amd as m
    for name, raw in snappy_raw.items():
        enc = oracle.encode(raw)
        assert len(enc) <= m.encode_bound(len(raw)) == oracle.encode_bound(len(raw))
        assert m.decode_size(enc) == len(raw) == oracle.decode_size(enc)
    with pytest.raises(m.LzfseError) as e:
        m.decode_size(b"bvx2")
    assert e.value.status == 8
    with pytest.raises(m.LzfseError) as e:
        m.decode_size(b"nope" + bytes(40))
    assert e.value.status == 2


def test_product_never_touches_oracle():
    """The product tree must not import, link or execute anything under oracle/."""
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "lzfse_rust_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("oracle/ (", "") or f == "codec.py" and False, (dirpath, f)


def test_status_strings(lib):
    assert lib.lzfse_mi_status_string(0) == b"ok"
    assert b"LMD payload" in lib.lzfse_mi_status_string(22)
