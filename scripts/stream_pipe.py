"""The reference's huge-file test at a smaller size (test/src/huge.rs:20-85: seq gen > lzfoo -encode > lzfoo -decode > seq test,
64 GB there): a masked Seq of GIB GiB is piped through `python -m lzfse_rust_amd.cli -encode --plain` and `... -decode`, both
streaming a window at a time, and checked as it comes out; the two codec processes' peak resident sets are reported (they
must not grow with the length of the stream).
    python scripts/stream_pipe.py [GiB]          (profiles/r03_stream_pipe.txt)"""
import os, subprocess, sys, threading, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = np.uint32(0x03030000)          # huge.rs:16: low entropy, plenty of matches
CHUNK = 16 << 20                      # bytes per generated piece


class Seq:
    """Seq::masked(Rng::default(), mask) in pieces: the LCG of test_kit (x <- x * 1103515245 + 12345 mod 2^32), vectorised by
    doubling in 32-bit arithmetic, the state carried from piece to piece"""

    def __init__(self):
        self.s = np.uint32(0)
        n = CHUNK // 4
        # jump tables for one piece: x_{k} = A_k x_0 + C_k
        a, c = np.uint32(1103515245), np.uint32(12345)
        self.A = np.empty(n, dtype=np.uint32); self.C = np.empty(n, dtype=np.uint32)
        self.A[0], self.C[0] = a, c
        filled, ak, ck = 1, a, c
        with np.errstate(over="ignore"):
            while filled < n:
                m = min(filled, n - filled)
                self.A[filled:filled + m] = self.A[:m] * ak
                self.C[filled:filled + m] = self.C[:m] * ak + ck
                filled += m
                ck = ak * ck + ck
                ak = ak * ak

    def piece(self):
        with np.errstate(over="ignore"):
            st = self.A * self.s + self.C
        self.s = st[-1]
        return (st & MASK).view(np.uint8)


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    total = int(gib * (1 << 30)) // CHUNK * CHUNK
    env = dict(os.environ, PYTHONPATH=ROOT)
    cli = [sys.executable, "-m", "lzfse_rust_amd.cli"]
    enc = subprocess.Popen(cli + ["-encode", "--plain"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    dec = subprocess.Popen(cli + ["-decode"], stdin=enc.stdout, stdout=subprocess.PIPE, env=env)
    enc.stdout.close()
    peak = {}

    def watch(name, p):
        best = 0
        while p.poll() is None:
            try:
                for line in open(f"/proc/{p.pid}/status"):
                    if line.startswith("VmHWM:"):
                        best = max(best, int(line.split()[1]))
            except OSError:
                pass
            time.sleep(0.2)
        peak[name] = best

    for name, p in (("encode", enc), ("decode", dec)):
        threading.Thread(target=watch, args=(name, p), daemon=True).start()

    def feed():
        g = Seq()
        for _ in range(total // CHUNK):
            enc.stdin.write(g.piece().tobytes())
        enc.stdin.close()

    t0 = time.time()
    th = threading.Thread(target=feed)
    th.start()
    g, got = Seq(), 0
    want = g.piece().tobytes(); at = 0
    while True:
        b = dec.stdout.read(1 << 20)
        if not b:
            break
        o = 0
        while o < len(b):
            k = min(len(b) - o, len(want) - at)
            assert b[o:o + k] == want[at:at + k], f"mismatch at byte {got + o}"
            o += k; at += k
            if at == len(want) and got + o < total:
                want = g.piece().tobytes(); at = 0
        got += len(b)
    th.join()
    assert enc.wait() == 0 and dec.wait() == 0
    assert got == total, (got, total)
    time.sleep(0.5)
    dt = time.time() - t0
    print(f"{total} bytes through encode | decode in {dt:.0f} s ({total / dt / 1e6:.0f} MB/s, generator and checker included); "
          f"peak resident set: encoder {peak.get('encode', 0) // 1024} MiB, decoder {peak.get('decode', 0) // 1024} MiB")


if __name__ == "__main__":
    main()
