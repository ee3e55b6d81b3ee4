"""lzfoo-like command line over the MI355X codec (reference: lzfoo/main.rs:30-194; same sub-commands and flags).

    python -m lzfse_rust_amd.cli -encode [-i FILE] [-o FILE] [-v] [--chunk BYTES] [--devices 0,1,..] [--plain]
    python -m lzfse_rust_amd.cli -decode [-i FILE] [-o FILE] [-v] [--devices 0,1,..]

Without -i / -o it reads standard input / writes standard output, like lzfoo. -encode writes this build's chunked
container ("LZMC": independent 4 MiB LZFSE streams, chunk c on device c mod g), or with --plain ONE
ordinary LZFSE stream: byte for byte what lzfoo itself writes (lzfoo/main.rs:89 encodes with LzfseRingEncoder::encode,
the ring / stream encoder's parse), streamed from the input to the output a window at a time in bounded memory, as
lzfoo's 512 KiB ring does. -decode takes either; a plain stream is decoded through LzfseRingDecoder::decode
(lzfoo/main.rs:104), streamed as well. -v prints lzfoo's statistics block (sizes, ratio, ns/B, MB/s of raw bytes) on
standard error.
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

    class _Counting:
        """the output side: counts what goes through"""

        def __init__(self, f):
            self.f, self.n = f, 0

        def write(self, b):
            self.f.write(b)
            self.n += len(b)

    class _Chained:
        """the bytes already taken from an input, then the input"""

        def __init__(self, head, f):
            self.head, self.f = head, f

        def read(self, n=-1):
            if not self.head:
                return self.f.read(n)
            if n is None or n < 0:
                out, self.head = self.head + self.f.read(), b""
                return out
            out, self.head = self.head[:n], self.head[n:]
            return out   # (a short read is allowed; the next call goes on)

    fin = open(a.input, "rb") if a.input else sys.stdin.buffer
    t0 = time.perf_counter()
    n_in = n_out = 0
    try:
        ctxs = [lz.Context(int(d)) for d in a.devices.split(",")]
        head = b""
        if mode == "decode":
            # container or plain stream? Exactly 4 bytes are taken (a pipe may hand over fewer per read, and not every input
            # can peek) and put back in front of the rest
            while len(head) < 4:
                more = fin.read(4 - len(head))
                if not more:
                    break
                head += more
            fin = _Chained(head, fin)
        streamed = (mode == "encode" and a.plain) or (mode == "decode" and head != b"LZMC")
        if streamed:
            # one LZFSE stream, reader -> writer (lzfoo/main.rs:86-107)
            fout = open(a.output, "wb") if a.output else sys.stdout.buffer
            sink = _Counting(fout)
            try:
                if mode == "encode":
                    n_in, n_out = lz.LzfseRingEncoder(context=ctxs[0]).encode(fin, sink)
                else:
                    n_in, n_out = lz.LzfseRingDecoder(context=ctxs[0]).decode(fin, sink)
                fout.flush()
            except BrokenPipeError:
                return 0
            finally:
                if a.output:
                    fout.close()
        else:
            data = fin.read()
            out = lz.encode_chunked(ctxs, data, a.chunk).tobytes() if mode == "encode" else lz.decode_chunked(ctxs, data).tobytes()
            n_in, n_out = len(data), len(out)
            if a.output:
                open(a.output, "wb").write(out)
            else:
                try:
                    sys.stdout.buffer.write(out)
                except BrokenPipeError:
                    return 0
    except lz.LzfseError as e:
        print(f"Error: {'Decode' if mode == 'decode' else 'Encode'}: {e}", file=sys.stderr)
        return 1
    if a.v:
        _stats(t0, n_in, n_out, a.input or "stdin", a.output or "stdout", mode)
    return 0


if __name__ == "__main__":
    sys.exit(main())
