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
    if lib.lzfse_mi_device_count() > 0:
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


def test_decode_headroom_is_the_most_a_damaged_block_can_over_produce(lib, oracle, snappy_raw, golden_dir):
    """lzfse_mi_decode_headroom (host code): 40 000 literals + 10 000 x 2 359 match bytes when the stream holds a bvx1 / bvx2
    block, 136 bytes per payload byte of a bvxn block, nothing for raw blocks (fse/constants.rs, vn/vn_core.rs)."""
    import ctypes as C
    import numpy as np

    def room(b):
        a = np.frombuffer(b, dtype=np.uint8)
        return lib.lzfse_mi_decode_headroom(a.ctypes.data if a.size else None, a.size)

    fse = 40000 + 10000 * 2359
    assert room(oracle.encode(snappy_raw["html"])) == fse
    assert room(open(os.path.join(golden_dir, "mutate", "vx1.lzfse"), "rb").read()) == fse
    assert room(open(os.path.join(golden_dir, "mutate", "raw.lzfse"), "rb").read()) == 0
    vxn = open(os.path.join(golden_dir, "mutate", "vxn.lzfse"), "rb").read()
    payload = int.from_bytes(vxn[8:12], "little")
    assert room(vxn) == 136 * (12 + payload)
    assert room(b"") == 0 and room(b"bvx$") == 0 and room(b"nope") == 0
    assert room(vxn[:20]) == 136 * 20      # cut short: what is there


def test_product_never_touches_oracle():
    """The product tree must not import, link or execute anything under oracle/: no source file under lzfse_rust_amd/
    so much as names it, and the built library neither needs the oracle's shared objects nor calls its entry points."""
    import shutil
    import subprocess
    from lzfse_rust_amd import build
    n_files = 0
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "lzfse_rust_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read().lower()
                assert "oracle" not in txt and "lzo_" not in txt, (dirpath, f)
                n_files += 1
    assert n_files >= 15
    build.build()
    for lib_path in (build.LIB_PATH, build.DIAG_LIB_PATH):
        dyn = subprocess.check_output([shutil.which("readelf") or "/usr/bin/readelf", "-d", lib_path], text=True)
        assert "oracle" not in dyn.lower()
        und = subprocess.check_output([shutil.which("nm") or "/usr/bin/nm", "-D", "--undefined-only", lib_path], text=True)
        assert "lzo_" not in und


def test_product_library_reads_no_environment_and_has_no_debug_hook():
    """Debug knobs live in the diagnostic build only (liblzfse_mi_diag.so): the shipped library imports no getenv and
    does not export the stage hook; the diagnostic one exports it."""
    import shutil
    import subprocess
    from lzfse_rust_amd import build
    build.build()
    nm = shutil.which("nm") or "/usr/bin/nm"
    und = subprocess.check_output([nm, "-D", "--undefined-only", build.LIB_PATH], text=True)
    assert "getenv" not in und
    prod = subprocess.check_output([nm, "-D", "--defined-only", build.LIB_PATH], text=True)
    diag = subprocess.check_output([nm, "-D", "--defined-only", build.DIAG_LIB_PATH], text=True)
    for hook in ("lzfse_mi_debug_candidates", "lzfse_mi_debug_last_lmds"):
        assert hook not in prod and hook in diag, hook
    for n in _declared_functions(os.path.join(ROOT, "include", "lzfse_mi.h")):
        assert n in prod and n in diag, n


def test_status_strings(lib):
    assert lib.lzfse_mi_status_string(0) == b"ok"
    assert b"LMD payload" in lib.lzfse_mi_status_string(22)


def test_small_size_classes_match_oracle(lib, oracle):
    """n <= 4096: raw / LZVN host path (frontend_bytes.rs:63-111) is bit-exact vs the oracle for every size
    on several byte distributions, and carries the reference's KATs (frontend_bytes.rs:455-510)."""
    import lzfse_rust_amd as m
    from oracle_py import rng_gen_vec, seq_masked
    rng = np.random.default_rng(2)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(200)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 200, size=2000))
    sources = [bytes(4096), text, seq_masked(7, 0x03030303, 4096), rng_gen_vec(1, 4096), b"ab" * 2048,
               bytes(range(256)) * 16, rng.integers(0, 3, size=4096, dtype=np.uint8).tobytes()]
    sizes = list(range(0, 80)) + list(range(250, 300)) + [511, 512, 1023, 1024, 2047, 2048, 4000, 4094, 4095, 4096]
    for s in sources:
        for n in sizes:
            got = m.encode_small(s[:n])
            assert got == oracle.encode(s[:n]), n
            assert oracle.decode(got) == s[:n]
    assert m.encode_small(b"test") == bytes([0x62, 0x76, 0x78, 0x2d, 4, 0, 0, 0, 0x74, 0x65, 0x73, 0x74, 0x62, 0x76, 0x78, 0x24])
    with pytest.raises(m.LzfseError):
        m.encode_small(bytes(4097))


def _build_c_driver(tmp_path):
    import subprocess
    from lzfse_rust_amd import build
    build.build()
    exe = str(tmp_path / "abi_driver")
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_driver.c"), "-L", libdir, "-llzfse_mi", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_header_is_c_and_c_driver_links(tmp_path, lib):
    """include/lzfse_mi.h compiles as C11 with -Wall -Wextra -Werror and a plain C caller links against the library.
    Without a GPU the driver must stop at lzfse_mi_create with LZFSE_MI_NO_DEVICE (exit code 3), never fall back."""
    import subprocess
    exe = _build_c_driver(tmp_path)
    rc = subprocess.run([exe], capture_output=True, text=True)
    # (the library's own device count: torch.cuda.is_available() has answered False in a process that had used HIP through the library first)
    assert rc.returncode == (0 if lib.lzfse_mi_device_count() > 0 else 3), (rc.returncode, rc.stdout, rc.stderr)


@pytest.mark.gpu
def test_c_driver_round_trip_on_gpu(tmp_path):
    """create / encode / decode_size / decode / error detail / batch / stream encode in windows / stream decode / destroy from C,
    no Python in the data path."""
    import subprocess
    exe = _build_c_driver(tmp_path)
    rc = subprocess.run([exe], capture_output=True, text=True)
    assert rc.returncode == 0, (rc.returncode, rc.stdout, rc.stderr)
    assert "abi driver ok" in rc.stdout
