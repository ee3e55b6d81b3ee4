"""Builds the in-tree HIP shared libraries (gfx950 only) with hipcc. No CPU fallback exists.

  liblzfse_mi.so        the product: reads no environment variable, exports only include/lzfse_mi.h
  liblzfse_mi_diag.so   the same sources with -DLZFSE_MI_DIAG: adds the LZFSE_MI_OPT_DIAG_* options and the stage-level
                        debug hook the test-suite uses to force code paths and to compare intermediate results
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "liblzfse_mi.so")
DIAG_LIB_PATH = os.path.join(_PKG, "liblzfse_mi_diag.so")
SOURCES = ["api.hip", "decode.hip", "encode.hip", "encode_match.hip", "encode_parse.hip", "host_small.cpp", "chunked.cpp", "stream.hip"]
HEADERS = ["common.h", "internal.h", "enc_common.h", os.path.join("..", "..", "include", "lzfse_mi.h")]
_OBJ = os.path.join(_PKG, "build")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X codec cannot be built")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(_CSRC, s))]


def needs_build(path=LIB_PATH):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(_CSRC, s) for s in _sources() + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _build_one(path, defines, verbose):
    cc = _hipcc()
    tag = "diag" if defines else "prod"
    os.makedirs(_OBJ, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fvisibility=hidden"] + defines

    def compile_one(src):
        obj = os.path.join(_OBJ, f"{tag}_{os.path.splitext(src)[0]}.o")
        cmd = [cc] + flags + ["-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", path + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(path + ".tmp", path)
    return path


def build(force=False, verbose=False):
    if force or needs_build(LIB_PATH):
        _build_one(LIB_PATH, [], verbose)
    if force or needs_build(DIAG_LIB_PATH):
        _build_one(DIAG_LIB_PATH, ["-DLZFSE_MI_DIAG"], verbose)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
