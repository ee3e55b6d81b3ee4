"""lzfoo-like command line over the MI355X codec (reference: lzfoo/main.rs:30-194; same sub-commands and flags).

    python -m lzfse_rust_amd.cli -encode [-i FILE] [-o FILE] [-v] [--chunk BYTES] [--devices 0,1,..] [--plain]
    python -m lzfse_rust_amd.cli -decode [-i FILE] [-o FILE] [-v] [--devices 0,1,..]

Without -i / -o it reads standard input / writes standard output, like lzfoo. -encode writes this build's chunked
container ("LZMC": independent 4 MiB LZFSE streams, chunk c on device c mod g), or with --plain ONE
ordinary LZFSE stream that any LZFSE decoder reads (the reference's own output format). -decode takes either.
-v prints lzfoo's statistics block (sizes, ratio, ns/B, MB/s of raw bytes) on standard error.
"""
import argparse
import sys
import time


def _stats(t0, n_in, n_out, name_in, name_out, mode):
    secs = time.perf_counter() - t0
    n_raw, n_payload = (n_in, n_out) if mode == "encode" else (n_out, n_in)
    if name_out == "stdout":
        print(file=sys.stderr)
    print(f"LZFSE {mode}", file=sys.stderr)
    print(f"Input: {name_in}", file=sys.stderr)
    print(f"Output: {name_out}", file=sys.stderr)
    print(f"Input size: {n_in} B", file=sys.stderr)
    print(f"Output size: {n_out} B", file=sys.stderr)
    print(f"Compression ratio: {n_raw / max(n_payload, 1):.3f}", file=sys.stderr)
    print(f"Speed: {1e9 * secs / max(n_raw, 1):.2f} ns/B, {n_raw / secs / 1024 / 1024:.2f} MB/s", file=sys.stderr)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] in ("-encode", "encode", "-decode", "decode"):
        mode = argv.pop(0).lstrip("-")
    else:
        print(__doc__, file=sys.stderr)
        return 2
    ap = argparse.ArgumentParser(prog=f"lzfse_rust_amd.cli -{mode}")
    ap.add_argument("-i", dest="input", metavar="FILE")
    ap.add_argument("-o", dest="output", metavar="FILE")
    ap.add_argument("-v", action="count", default=0)
    ap.add_argument("--devices", default="0", help="comma separated HIP device indices (chunk c -> device c mod g)")
    if mode == "encode":
        ap.add_argument("--chunk", type=int, default=0, help="chunk size in bytes (default 4 MiB)")
        ap.add_argument("--plain", action="store_true", help="one ordinary LZFSE stream instead of the chunked container")
    a = ap.parse_args(argv)

    import lzfse_rust_amd as lz
    data = open(a.input, "rb").read() if a.input else sys.stdin.buffer.read()
    t0 = time.perf_counter()
    try:
        ctxs = [lz.Context(int(d)) for d in a.devices.split(",")]
        if mode == "encode":
            if a.plain:
                dst = bytearray()
                lz.LzfseEncoder(context=ctxs[0]).encode_bytes(data, dst)
                out = bytes(dst)
            else:
                out = lz.encode_chunked(ctxs, data, a.chunk).tobytes()
        elif data[:4] == b"LZMC":
            out = lz.decode_chunked(ctxs, data).tobytes()
        else:
            dst = bytearray()
            lz.LzfseDecoder(context=ctxs[0]).decode_bytes(data, dst)
            out = bytes(dst)
    except lz.LzfseError as e:
        print(f"Error: {'Decode' if mode == 'decode' else 'Encode'}: {e}", file=sys.stderr)
        return 1
    if a.output:
        open(a.output, "wb").write(out)
    else:
        try:
            sys.stdout.buffer.write(out)
        except BrokenPipeError:
            return 0
    if a.v:
        _stats(t0, len(data), len(out), a.input or "stdin", a.output or "stdout", mode)
    return 0


if __name__ == "__main__":
    sys.exit(main())
