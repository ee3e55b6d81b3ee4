#!/usr/bin/env python3
"""Headline benchmark: encode+decode MB/s on the Snappy corpus (BASELINE.json), per GPU and aggregated over --gpus N
ranks. One process per GPU; `python bench.py --gpus N` starts its N workers itself (or runs as one rank under
torchrun); independent streams are sharded across ranks, no collective on the data path (RCCL only carries the timing
barrier / max-reduce and the result metadata).

A step = one pass of the hot path over one batch: encode (raw -> streams) then decode (streams -> raw), inputs and
outputs resident in HBM. value = raw bytes through the encode+decode round trip per second (10^6 bytes/s), whole job.

Workloads (--workload):
  snappy    the 12 Snappy files x --replicas copies per GPU as independent streams (the metric's config; weak scaling)
  chunks1g  ONE 1 GiB input cut into 256 x 4 MiB streams, chunk c -> rank c mod N (BASELINE config 5; strong scaling);
            every chunk stream is compared (SHA-256) with the list rank 0 produces alone
  text64m   64 MiB of text as ONE stream per GPU (configs 2 / 3)
  chunks4m  256 MiB of text as 64 x 4 MiB streams per GPU

Prints ONE JSON line (rank 0). Besides the contract's keys: roofline (dominant kernel, per launch over the timed
region + one exclusive unsplit pass), cpu_baseline (the C restatement of lzfse_rust's CPU path on this box's cores),
pcie_inclusive, copy_peak, encode/decode splits.
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_corpus_streams():
    g = os.path.join(ROOT, "tests", "golden", "snappy")
    names = sorted(os.path.basename(f)[:-6] for f in glob.glob(os.path.join(g, "*.lzfse")))
    return names, [open(os.path.join(g, n + ".lzfse"), "rb").read() for n in names], \
        [open(os.path.join(g, n + ".hash"), "rb").read() for n in names]


def synth_text(n_bytes, seed=1):
    """Deterministic enwik-style text: order-1 word chain over a fixed
    4096-word vocabulary drawn with the reference's LCG (test_kit/src/rng.rs:14-17)."""
    s = seed & 0xFFFFFFFF

    def gen():
        nonlocal s
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        return s >> 8
    vocab = []
    for _ in range(4096):
        ln = 2 + gen() % 9
        vocab.append(bytes(97 + gen() % 26 for _ in range(ln)))
    rng = np.random.default_rng(seed)
    # zipf-ish word choice with a short-range repeat bias, vectorised
    n_words = n_bytes // 5 + 16
    ranks = np.minimum((rng.pareto(1.1, size=n_words) * 12).astype(np.int64), 4095)
    rep = rng.random(n_words) < 0.08
    back = rng.integers(1, 64, size=n_words)
    idx = np.arange(n_words)
    src = np.where(rep & (idx >= back), idx - back, idx)
    ranks = ranks[src]
    seps = np.where(rng.random(n_words) < 0.07, 1, 0)
    parts = []
    total = 0
    for r, sp in zip(ranks.tolist(), seps.tolist()):
        w = vocab[r]
        parts.append(w)
        parts.append(b". " if sp else b" ")
        total += len(w) + 1 + sp
        if total >= n_bytes:
            break
    return b"".join(parts)[:n_bytes]


def corpus_1g(seed=1):
    """1 GiB = 16 copies of 64 MiB of synthetic text, each copy perturbed in one byte out of 251
    so that the copies do not match each other. The same bytes on every rank."""
    base = np.frombuffer(synth_text(64 << 20, seed=seed), dtype=np.uint8)
    out = np.empty(16 * base.size, dtype=np.uint8)
    for c in range(16):
        a = out[c * base.size:(c + 1) * base.size]
        a[:] = base
        a[c % 251::251] ^= np.uint8(1 + c)
    return out


class GpuCodec:
    """The product's host-pointer batch API in the shape lzfse_rust_amd.sharding wants."""

    def __init__(self, ctx, max_batch=64):
        self.ctx, self.max_batch = ctx, max_batch

    def encode_batch(self, raws):
        out = []
        for i in range(0, len(raws), self.max_batch):
            o, st = self.ctx.encode_batch(raws[i:i + self.max_batch])
            assert all(s == 0 for s in st), st
            out += [x.tobytes() for x in o]
        return out

    def decode_batch(self, encs, raw_lens):
        out = []
        for i in range(0, len(encs), self.max_batch):
            o, st = self.ctx.decode_batch(encs[i:i + self.max_batch], caps=raw_lens[i:i + self.max_batch])
            assert all(s == 0 for s in st), st
            out += [x.tobytes() for x in o]
        return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell: start one fresh worker per GPU (nothing in this process has touched
    the GPU or torch.cuda) and hand back the first non-zero exit code. Workers rendezvous on 127.0.0.1."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rcs = [p.wait() for p in procs]
    return next((rc for rc in rcs if rc), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicas", type=int, default=256, help="copies of the 12-file corpus per GPU (256: 3072 streams, 752 MB)")
    ap.add_argument("--workload", default="snappy", choices=["snappy", "text64m", "chunks4m", "chunks1g"])
    ap.add_argument("--emulate-world", type=int, default=0, metavar="N",
                    help="chunks1g only: run rank 0's shard of an N-way split on ONE GPU (what one rank of an N-GPU strong-scaling run does)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exclusive pass, PCIe-inclusive and copy-peak measurements")
    ap.add_argument("--lanes", type=int, default=0, help="sub-batches run side by side per call (0: library default, 1: unsplit)")
    ap.add_argument("--decode-pipe", type=int, default=0, help="LZFSE_MI_OPT_DECODE_PIPE (0: by the batch's shape, 1: never)")
    ap.add_argument("--stagger", action="store_true", help="encode lanes start one after the other instead of together")
    ap.add_argument("--skip-verify", action="store_true", help=argparse.SUPPRESS)  # timing of deliberately broken ablation builds
    ap.add_argument("--per-file", type=int, default=0, metavar="R",
                    help="instead of the headline run: Criterion-style table, each Snappy file alone as a batch of R copies")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` from a plain shell, or "
                 f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import lzfse_rust_amd as lz
    from lzfse_rust_amd import sharding
    ctx = lz.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.decode_pipe:
        ctx.set_option("decode_pipe", args.decode_pipe)
    if args.lanes:
        ctx.set_option("encode_lanes", args.lanes)
        ctx.set_option("decode_lanes", args.lanes)
    if args.stagger:
        ctx.set_option("stagger", 1)

    names, fixture_streams, hashes = load_corpus_streams()
    if args.per_file:
        per_file_table(ctx, torch, dev, names, fixture_streams, args.per_file)
        return

    # ---- build the per-GPU batch (untimed) ----
    scaling = "weak"
    if args.workload == "snappy":
        raws_np, st = ctx.decode_batch(fixture_streams)  # product decoder recovers the raw corpus
        assert all(s == 0 for s in st)
        raws = [r.tobytes() for r in raws_np]
        for r, h, n in zip(raws, hashes, names):
            assert hashlib.sha256(r).digest() == h, n
        batch_raw = raws * args.replicas
        workload = f"snappy corpus (12 files, {sum(map(len, raws))} B) x {args.replicas} replicas per GPU as independent streams, encode+decode"
    elif args.workload == "text64m":
        batch_raw = [synth_text(64 << 20, seed=1 + rank)]
        workload = "64 MiB synthetic enwik-style text, ONE stream per GPU, encode+decode"
    elif args.workload == "chunks1g":
        # strong scaling: the same 1 GiB on every rank, chunk c -> rank c mod world
        scaling = "strong"
        data = corpus_1g(seed=1)
        bounds = sharding.chunk_bounds(data.size, sharding.CHUNK_BYTES)
        emu = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
        mine = sharding.shard(len(bounds), rank, emu or world)
        batch_raw = [data[o:o + n].tobytes() for o, n in (bounds[c] for c in mine)]
        workload = (f"ONE 1 GiB input (16 perturbed copies of 64 MiB synthetic text) cut into {len(bounds)} independent 4 MiB "
                    f"streams, chunk c -> rank c mod {emu or world}, encode+decode")
        if emu:
            workload += f"; EMULATED: only rank 0's shard of a {emu}-way split ({len(mine)} streams) on one GPU"
        # untimed: the sharded result equals what ONE encoder makes of the same chunks (per-chunk SHA-256)
        report = sharding.process_shard(data, sharding.CHUNK_BYTES, rank, world, GpuCodec(ctx))
        reports = sharding.gather_reports(report, world, dist)
        if rank == 0:
            merged = sharding.merge_reports(reports, len(bounds))
            if world > 1:
                sharding.check_against(merged, sharding.process_shard(data, sharding.CHUNK_BYTES, 0, 1, GpuCodec(ctx)))
        # one PROCESS over every visible device (what a Rust caller of lzfse_mi_encode_chunked does; untimed, rank 0): the frame of
        # the first 64 MiB over all devices equals the one a single context makes, and decodes back on all of them
        chunked_devices = lz.device_count()
        if rank == 0 and chunked_devices > 1:
            cx = [ctx] + [lz.Context(k) for k in range(chunked_devices) if k != local_rank]
            part = data[:64 << 20]
            frame = lz.encode_chunked(cx, part, sharding.CHUNK_BYTES)
            assert frame.tobytes() == lz.encode_chunked(cx[:1], part, sharding.CHUNK_BYTES).tobytes()
            assert np.array_equal(np.asarray(lz.decode_chunked(cx, frame)), part)
            for c in cx[1:]:
                c.close()
        # ... and what that caller gets on ONE device: the whole 1 GiB through lzfse_mi_encode_chunked / _decode_chunked (host
        # pointers in and out, the frame assembled in the caller's buffer; best of 2, N = 1 only)
        chunked_rate = None
        if rank == 0 and world == 1 and not emu and not args.no_extras:
            best_e = best_d = 1e9
            frame = None
            for _ in range(2):
                t0 = time.perf_counter()
                frame = lz.encode_chunked([ctx], data, sharding.CHUNK_BYTES)
                best_e = min(best_e, time.perf_counter() - t0)
            for _ in range(2):
                t0 = time.perf_counter()
                back = lz.decode_chunked([ctx], frame)
                best_d = min(best_d, time.perf_counter() - t0)
            assert np.array_equal(np.asarray(back), data)
            chunked_rate = {"encode_GBps": round(data.size / best_e / 1e9, 2), "decode_GBps": round(data.size / best_d / 1e9, 2),
                            "frame_bytes": int(frame.size),
                            "what": "the whole 1 GiB through lzfse_mi_encode_chunked / lzfse_mi_decode_chunked on one context (host memory in and "
                                    "out, 256 chunks of 4 MiB framed as LZMC; the Python binding allocates the destination; best of 2)"}
            del back, frame
        del data
    else:
        t = synth_text(256 << 20, seed=1 + rank)
        batch_raw = [t[i:i + (4 << 20)] for i in range(0, len(t), 4 << 20)]
        workload = "256 MiB synthetic text cut into 64 independent 4 MiB streams per GPU, encode+decode"
    n_streams = len(batch_raw)
    raw_total = sum(len(r) for r in batch_raw)

    B = DeviceBatch(torch, dev, lz, batch_raw)
    # encode once (untimed) to learn stream sizes and to verify the round trip bit-exactly
    enc_len, est = ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
    assert (est == 0).all(), est
    dec_len, dst_ = ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
    assert args.skip_verify or (dst_ == 0).all(), dst_
    if not args.skip_verify:
        assert (dec_len == B.raw_len).all()
        assert torch.equal(B.d_dec[:B.raw_padded], B.d_raw[:B.raw_padded]), "round trip mismatch"
    comp_total = int(enc_len.sum())

    def step(timed):
        kern = {}
        t0 = time.perf_counter()
        ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
        torch.cuda.synchronize()
        t_e = time.perf_counter() - t0
        if timed:
            kern.update(ctx.timings())
        t0 = time.perf_counter()
        ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
        torch.cuda.synchronize()
        t_d = time.perf_counter() - t0
        if timed:
            kern.update(ctx.timings())
        return t_e, t_d, kern

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.enable_timing(True)
    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    te = td = 0.0
    kern_ms, kern_n = {}, {}
    for _ in range(args.steps):
        a, b, k = step(True)
        te += a
        td += b
        for name, (ms, n) in k.items():
            kern_ms[name] = kern_ms.get(name, 0.0) + ms
            kern_n[name] = kern_n.get(name, 0) + n
    barrier()
    elapsed = time.perf_counter() - t_start

    stats = torch.tensor([elapsed, te, td], dtype=torch.float64, device=dev)
    totals = torch.tensor([raw_total, comp_total, n_streams], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    elapsed, te, td = stats.tolist()
    raw_all, comp_all, streams_all = (int(x) for x in totals.tolist())

    failed = False
    if rank == 0:
        value = raw_all * args.steps / elapsed / 1e6
        kx = None
        if not args.no_extras:
            # one UNSPLIT pass (one launch per kernel, nothing co-scheduled): exclusive kernel durations
            ctx.set_option("encode_lanes", 1)
            ctx.set_option("decode_lanes", 1)
            step(False)
            _, _, kx = step(True)
            ctx.set_option("encode_lanes", args.lanes)
            ctx.set_option("decode_lanes", args.lanes)
        # dominant kernel: by exclusive device time when that pass ran (the sub-batches of the timed region run side by
        # side, so a kernel that merely waits for its share of the chip next to another one accumulates the most event
        # time there), else by the accumulated event time of the timed region
        common = [k for k in kern_ms if kx and k in kx]
        dom = max(common, key=lambda k: kx[k][0]) if common else max(kern_ms, key=kern_ms.get)
        dom_avg_ms = kern_ms[dom] / max(kern_n[dom], 1)
        is_dec = dom.startswith("dec")
        # B_dec = compressed_in + raw_out ; B_enc = raw_in + compressed_out, per step (this rank); a batch call may be cut
        # into sub-batches that run side by side (split batches), so one launch covers bytes-per-step / launches
        alg_bytes = int((comp_total + raw_total) * args.steps / max(kern_n[dom], 1))
        achieved = alg_bytes / (dom_avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_avg_ms, 4),
                "direction": "decode" if is_dec else "encode",
                # the launches of one step overlap on the device (sub-batches side by side): a per-launch duration under
                # co-scheduling is not an exclusive kernel time; see exclusive_* below for one unsplit pass
                "launches_per_step": round(kern_n[dom] / args.steps, 2)}
        roof["traffic"], enc_traffic = pmc_traffic(dom, args, comp_total + raw_total)
        roof["traffic_is"] = "2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 read-counter correction of MI355X_MICROARCH.md applied)"
        if enc_traffic:
            roof["encoder_traffic"] = enc_traffic
        out = {
            "metric": "encode+decode MB/s on Snappy corpus",
            "value": round(value, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if args.workload != "snappy" else "snappy corpus fixtures replicated (weights n/a)",
            "config": {"workload": workload, "streams_all_gpus": streams_all, "raw_bytes_per_step_all_gpus": raw_all,
                       "compressed_bytes_per_step_all_gpus": comp_all, "streams_rank0": n_streams, "raw_bytes_rank0": raw_total},
            "encode_MBps": round(raw_all * args.steps / te / 1e6, 2),
            "decode_MBps": round(raw_all * args.steps / td / 1e6, 2),
            "kernel_ms_per_step": {k: round(v / args.steps, 4) for k, v in sorted(kern_ms.items())},
            "roofline": roof,
        }
        if args.workload == "chunks1g":
            out["chunked_over_devices"] = chunked_devices   # (> 1: lzfse_mi_encode_chunked ran over that many devices from rank 0, checked)
            if chunked_rate:
                out["chunked_one_device"] = chunked_rate
        if kx is not None:
            if dom in kx:   # (an unsplit call may take another LZ path than its sub-batches did)
                ex_ms = kx[dom][0] / max(kx[dom][1], 1)
                roof["exclusive_launch_ms"] = round(ex_ms, 4)
                roof["exclusive_achieved"] = round((comp_total + raw_total) / (ex_ms * 1e-3) / 1e9, 3)
                roof["exclusive_frac"] = round(roof["exclusive_achieved"] / HBM_PEAK_GBPS, 6)
            out["exclusive_kernel_ms"] = {k: round(v[0], 4) for k, v in sorted(kx.items())}
            out["copy_peak"] = copy_peak(torch, dev)
            out["pcie_inclusive"] = pcie_inclusive(ctx, lz, batch_raw)
            if args.workload == "snappy":
                ctx.enable_timing(False)
                out["html"] = html_rows(ctx, torch, dev, lz, raws[names.index("html")])
                out["per_file"] = per_file_compact(ctx, torch, dev, lz, names, raws)
                out["stream"] = stream_rows(ctx, lz)
        if not args.no_cpu_baseline:
            sample = batch_raw[:12] if args.workload == "snappy" else [batch_raw[0][:8 << 20]]
            out["cpu_baseline"] = cpu_baseline(sample)
        # the pipelined LZ stage of decode was never given up (a context that does so silently decodes 2-3 x slower)
        out["pipe_refusals"] = ctx.pipe_refusals()
        print(json.dumps(out), flush=True)
        if out["pipe_refusals"] != 0:   # (the record is out; the exit code says that it is not the configuration wanted)
            print("bench.py: the pipelined LZ kernel was switched off on this device (pipe_refusals != 0)", file=sys.stderr)
            failed = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


class DeviceBatch:
    """Streams of a batch laid out 256-byte aligned in device buffers (inputs, encoded, decoded)."""

    def __init__(self, torch, dev, lz, batch_raw):
        def layout(lens, align=256):
            off, o = [], 0
            for n in lens:
                off.append(o)
                o += (n + align - 1) // align * align
            return np.array(off, dtype=np.uint64), o
        self.raw_len = np.array([len(r) for r in batch_raw], dtype=np.uint64)
        self.raw_off, self.raw_padded = layout(self.raw_len)
        self.enc_cap = np.array([lz.encode_bound(int(n)) for n in self.raw_len], dtype=np.uint64)
        self.enc_off, enc_padded = layout(self.enc_cap)
        h_raw = np.zeros(self.raw_padded + 256, dtype=np.uint8)
        for r, o in zip(batch_raw, self.raw_off):
            h_raw[int(o):int(o) + len(r)] = np.frombuffer(r, dtype=np.uint8)
        self.d_raw = torch.from_numpy(h_raw).to(dev)
        self.d_enc = torch.zeros(enc_padded + 256, dtype=torch.uint8, device=dev)
        self.d_dec = torch.zeros(self.raw_padded + 256, dtype=torch.uint8, device=dev)


def kernel_source_sha():
    """Identity of the kernels a PMC pass was taken on: SHA-256 over the code lines of the HIP sources (the GPU box has
    no .git)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lzfse_rust_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        # code only: `//` comments and blank lines do not change a kernel
        for line in open(os.path.join(d, f), "r", errors="replace"):
            code = line.split("//", 1)[0].strip()
            if code:
                h.update(code.encode())
                h.update(b"\n")
    return h.hexdigest()[:16]


def pmc_files():
    """profiles/rNN_pmc_traffic.json, newest round first (the one stamped with this build's kernel sources is the one used)"""
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True)


def pmc_traffic(dom, args, b_enc=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same default command, profiles/rNN_pmc_traffic.json), CORRECTED as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950: 2 x FETCH_SIZE + WRITE_SIZE (the read counter tallies 128-byte requests at 64 bytes;
    checked on this build's own mandatory coalesced reads). Only reported when that file was taken
    on exactly the kernel sources of this build (source_sha) and for the default workload; otherwise null. Returns
    (bytes per launch of `dom`, {whole-encoder figures}) ."""
    try:
        if args.workload != "snappy" or args.replicas != 256 or args.lanes:
            return None, None
        sha = kernel_source_sha()
        pm = next((m for m in (json.load(open(f)) for f in pmc_files()) if m.get("source_sha") == sha), None)
        if pm is None:
            return None, None
        ks = pm["kernels"]

        def per_launch(prefix):
            key = [k for k in ks if k.startswith(prefix + "_kernel")]
            if not key:
                return None
            return int(sum((2 * ks[k]["fetch_bytes_raw"] + ks[k]["write_bytes"]) * ks[k]["launches"] for k in key) /
                       sum(ks[k]["launches"] for k in key))
        enc = None
        if b_enc:
            # every encode kernel, bytes per STEP (launches per step come from the file's own launch counts: the profiled
            # command makes pm["steps"] encode calls)
            steps = pm.get("encode_calls", 0)
            if steps:
                tot = sum((2 * v["fetch_bytes_raw"] + v["write_bytes"]) * v["launches"] for k, v in ks.items() if k.startswith("enc_")) / steps
                enc = {"hbm_bytes_per_step": int(tot), "over_algorithmic": round(tot / b_enc, 2)}
        return per_launch(dom), enc
    except Exception:
        return None, None


def copy_peak(torch, dev):
    """Measured device copy rate (read + write bytes / time) beside the 8 TB/s spec."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return {"GBps": round(2 * n / (ms * 1e-3) / 1e9, 1), "what": "torch uint8 device-to-device copy of 1 GiB, read + write bytes"}


def pcie_inclusive(ctx, lz, batch_raw, n_streams=384):
    """The host-pointer entry points (what the Rust shim binds): pageable host buffers in, pinned staging, one H2D and one
    D2H per batch, results back in host memory. Never `value`."""
    import ctypes as C
    ctx.enable_timing(False)   # (stage timings are per context: a large host call is not cut in two while they are collected)
    sample = batch_raw[:n_streams]
    n = len(sample)
    raw = sum(len(r) for r in sample)
    arrs = [np.frombuffer(r, dtype=np.uint8) for r in sample]
    L = ctx._lib

    def call(fn, srcs, caps):
        outs = [np.empty(int(c), dtype=np.uint8) for c in caps]
        sp = (C.c_void_p * n)(*[a.ctypes.data for a in srcs])
        sl = (C.c_size_t * n)(*[a.size for a in srcs])
        dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        dc = (C.c_size_t * n)(*[int(c) for c in caps])
        ol = (C.c_size_t * n)()
        st = (C.c_int * n)()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = fn(ctx._h, n, sp, sl, dp, dc, ol, st)
            best = min(best, time.perf_counter() - t0)
        assert rc == 0 and all(s == 0 for s in st)
        return [o[:ol[i]] for i, o in enumerate(outs)], best
    encs, t_e = call(L.lzfse_mi_encode_batch, arrs, [lz.encode_bound(a.size) for a in arrs])
    _, t_d = call(L.lzfse_mi_decode_batch, encs, [a.size for a in arrs])
    return {"encode_MBps": round(raw / t_e / 1e6, 1), "decode_MBps": round(raw / t_d / 1e6, 1),
            "sample": f"{n} streams, {raw} raw bytes through lzfse_mi_encode_batch / _decode_batch (host pointers), best of 3"}


def cpu_quota():
    """CPUs this process may use at once (the cgroup's quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())        # cgroup v1
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(sample):
    """The oracle (C restatement of lzfse_rust's CPU path; the reference itself cannot be built: no Rust toolchain)
    timed on this box's host cores by oracle/lzo_bench.c: pthreads that loop over their own streams for the whole
    budget, no interpreter in the loop -- one thread (comparable with the reference README's `rust` column), then one
    thread per core over independent streams. About 20 s in all."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import Oracle
    o = Oracle("liblzfse_oracle_native.so")
    encs = [o.encode(r) for r in sample]   # (also initialises the oracle's tables before any thread runs)
    nbytes = sum(len(r) for r in sample)
    n = len(sample)
    ra = [np.frombuffer(r, dtype=np.uint8) for r in sample]
    ea = [np.frombuffer(e, dtype=np.uint8) for e in encs]
    rp = (C.c_void_p * n)(*[a.ctypes.data for a in ra])
    rl = (C.c_size_t * n)(*[a.size for a in ra])
    ep = (C.c_void_p * n)(*[a.ctypes.data for a in ea])
    el = (C.c_size_t * n)(*[a.size for a in ea])
    fn = o.lib.lzo_bench_threads
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int,
                   C.POINTER(C.c_double), C.POINTER(C.c_uint64)]

    def run(threads, budget):
        rates = []
        for decode in (0, 1):
            mbps, tot = C.c_double(0), C.c_uint64(0)
            st = fn(rp, rl, ep, el, n, threads, budget, decode, C.byref(mbps), C.byref(tot))
            assert st == 0, st
            rates.append(mbps.value)
        e, d = rates
        return 1.0 / (1.0 / e + 1.0 / d), e, d     # encode + decode of the same bytes, one after the other
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    visible = cores
    quota = cpu_quota()
    if quota and quota < cores:
        cores = max(1, int(quota + 0.5))   # the box's CPU share: more threads than that only take turns
    v1, e1, d1 = run(1, 4.0)
    vn, en, dn = run(cores, 5.0) if cores > 1 else (v1, e1, d1)
    return {"value": round(vn, 2), "unit": "MB/s", "cores": cores, "kind": "port",
            "encode_MBps": round(en, 2), "decode_MBps": round(dn, 2),
            "single_thread": {"value": round(v1, 2), "encode_MBps": round(e1, 2), "decode_MBps": round(d1, 2)},
            "threads_over_single": {"encode": round(en / e1, 1), "decode": round(dn / d1, 1)},
            "cpus_visible": visible, "cpu_quota": quota,
            "sample": f"{n} stream(s), {nbytes} raw bytes; per direction ~4 s on 1 thread and ~5 s on {cores} pthreads (cores = threads used: "
                      f"the cgroup's CPU quota when there is one, else the CPUs visible), every "
                      "thread looping over its own streams for the whole budget (oracle/lzo_bench.c, no Python in the "
                      "loop); value = 1 / (1/encode + 1/decode); gcc -O3 -march=native C restatement of lzfse_rust's slice "
                      "path (oracle/)"}


# README.md:155-176 of the reference: Criterion, i5-2500K, single thread, MiB/s of raw bytes, column `rust` (decode, encode)
README_I5_2500K = {
    "html": (945.7, 118.9), "urls.10K": (552.5, 74.2), "fireworks.jpeg": (355.0, 61.5), "paper-100k.pdf": (429.0, 63.9),
    "html_x_4": (3174.4, 457.2), "alice29.txt": (344.7, 55.1), "asyoulik.txt": (319.7, 51.2), "lcet10.txt": (371.0, 58.5),
    "plrabn12.txt": (304.0, 49.7), "geo.protodata": (1254.1, 140.9), "kppkn.gtb": (425.4, 74.8),
}


def file_rates(ctx, torch, dev, lz, raw, R, samples=5):
    """One file alone as a batch of R independent copies resident in HBM: wall time of the encode and of the decode call,
    2 warm-up + `samples` samples (the protocol of per_file_table). Returns (B, enc_len, te[], td[])."""
    B = DeviceBatch(torch, dev, lz, [raw] * R)
    enc_len, est = ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
    assert (est == 0).all()
    te, td = [], []
    import gc
    gc_was = gc.isenabled() and not os.environ.get("BENCH_KEEP_GC")
    if gc_was:
        gc.disable()       # (as timeit does: a cyclic collection inside a 1 ms sample is the interpreter's time, not the call's)
    for _ in range(samples + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, dst_ = ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert (dst_ == 0).all()
        te.append(t1 - t0)
        td.append(t2 - t1)
    if gc_was:
        gc.enable()
    assert torch.equal(B.d_dec[:B.raw_padded], B.d_raw[:B.raw_padded])
    return B, enc_len, np.array(te[2:]), np.array(td[2:])


def html_rows(ctx, torch, dev, lz, raw):
    """The file the north-star target is phrased on (README.md:155,166 of the reference, bench/src/bench.rs:181-193), by
    itself: R = 256 and R = 16 copies per call, so that a driver record carries it."""
    rows = {}
    for R in (256, 16):
        _, enc_len, te, td = file_rates(ctx, torch, dev, lz, raw, R, samples=20 if R == 256 else 5)
        rows[f"x{R}"] = {"encode_GBps": round(len(raw) * R / te.mean() / 1e9, 2), "decode_GBps": round(len(raw) * R / td.mean() / 1e9, 2),
                         "encode_sd_pct": round(100 * te.std() / te.mean(), 1), "decode_sd_pct": round(100 * td.std() / td.mean(), 1)}
    rows["what"] = (f"data/snappy/html ({len(raw)} B) alone as a batch of R independent copies resident in HBM, wall time of the "
                    "batch call, 2 warm-up + 20 (x256) / 5 (x16) samples")
    return rows


def stream_rows(ctx, lz, mib=128):
    """SURVEY section 8f rank 3 inside the default line: LzfseRingEncoder::encode / LzfseRingDecoder::decode over a reader and a
    writer (host memory on both sides, 1 MiB reads), synthetic text, by the window (scripts/stream_bench.py has all the rows)."""
    import io

    class Count:
        def __init__(self):
            self.n = 0

        def write(self, b):
            self.n += len(b)

    raw = bytes(synth_text(mib << 20))
    rows = {}
    enc = None
    for window in (64 << 20, 16 << 20):
        best_e = best_d = 1e9
        for _ in range(2):
            out = bytearray()
            w = lz.LzfseRingEncoder(context=ctx, window=window, read_size=1 << 20)
            t0 = time.perf_counter()
            u, v = w.encode(io.BytesIO(raw), _Appender(out))
            best_e = min(best_e, time.perf_counter() - t0)
            assert u == len(raw)
            enc = bytes(out)
        for _ in range(2):
            sink = Count()
            t0 = time.perf_counter()
            u, v = lz.LzfseRingDecoder(context=ctx, window=window, read_size=1 << 20).decode(io.BytesIO(enc), sink)
            best_d = min(best_d, time.perf_counter() - t0)
            assert (u, v, sink.n) == (len(enc), len(raw), len(raw))
        rows[f"window_{window >> 20}MiB"] = {"encode_GBps": round(len(raw) / best_e / 1e9, 2), "decode_GBps": round(len(raw) / best_d / 1e9, 2)}
    rows["what"] = (f"{mib} MiB of synthetic text through LzfseRingEncoder.encode(reader, writer) / LzfseRingDecoder.decode(reader, writer) "
                    "(the Python mirrors over lzfse_mi_estream_* / _dstream_*; host buffers, 1 MiB reads, best of 2); the ring parse's bytes")
    return rows


class _Appender:
    def __init__(self, out):
        self.out = out

    def write(self, b):
        self.out += b


def per_file_compact(ctx, torch, dev, lz, names, raws, R=256, samples=20):
    """BASELINE config 4 inside the default line: every Snappy file by itself as a batch of R copies resident in HBM, Criterion's
    sample count (bench/src/bench.rs:5-6,279-283: 20 samples), [encode GB/s, decode GB/s] from the median wall time of the batch
    calls. The full table (sd, CPU port column, bit-exact check per file) is --per-file R."""
    rows = {}
    for name, raw in zip(names, raws):
        _, _, te, td = file_rates(ctx, torch, dev, lz, raw, R, samples=samples)
        rows[name] = [round(len(raw) * R / float(np.median(te)) / 1e9, 2), round(len(raw) * R / float(np.median(td)) / 1e9, 2)]
    rows["what"] = (f"each file alone as a batch of {R} independent copies resident in HBM, 2 warm-up + {samples} samples, "
                    "[encode GB/s, decode GB/s] of the MEDIAN wall time of the batch call (one stalled sample of twenty moves a "
                    "mean by a factor of two on a 3 ms call), round trip checked")
    return rows


def per_file_table(ctx, torch, dev, names, fixture_streams, R):
    """BASELINE config 4: each Snappy file by itself (bench/src/bench.rs:181-193,279-283 decode/encode pairs), as a batch of
    R independent copies resident in HBM; beside the published i5-2500K numbers and the CPU port on this box (1 thread)."""
    import lzfse_rust_amd as lz
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import Oracle
    o = Oracle("liblzfse_oracle_native.so")
    raws_np, st = ctx.decode_batch(fixture_streams)
    assert all(s == 0 for s in st)
    rows = []
    for name, r in zip(names, raws_np):
        raw = r.tobytes()
        B, enc_len, te, td = file_rates(ctx, torch, dev, lz, raw, R, samples=20)
        want = o.encode(raw)
        got = B.d_enc[int(B.enc_off[0]):int(B.enc_off[0]) + int(enc_len[0])].cpu().numpy().tobytes()
        assert got == want, name   # bit-exact vs the CPU port
        # CPU port, one thread, ~0.5 s per direction
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.5:
            o.encode(raw)
            k += 1
        ce = time.perf_counter() - t0
        t0 = time.perf_counter()
        j = 0
        while time.perf_counter() - t0 < 0.5:
            o.decode(want, cap=len(raw), as_array=True)
            j += 1
        cd = time.perf_counter() - t0
        mib = 1 << 20
        pub = README_I5_2500K.get(name)
        rows.append({
            "file": name, "raw_bytes": len(raw), "compressed_bytes": len(want), "copies": R,
            "gpu_encode_MiBps": round(len(raw) * R / te.mean() / mib, 1), "gpu_encode_sd_pct": round(100 * te.std() / te.mean(), 1),
            "gpu_decode_MiBps": round(len(raw) * R / td.mean() / mib, 1), "gpu_decode_sd_pct": round(100 * td.std() / td.mean(), 1),
            "gpu_encode_GBps": round(len(raw) * R / float(np.median(te)) / 1e9, 2), "gpu_decode_GBps": round(len(raw) * R / float(np.median(td)) / 1e9, 2),
            "gpu_encode_ms_median_max": [round(float(np.median(te)) * 1e3, 3), round(float(te.max()) * 1e3, 3)],
            "gpu_decode_ms_median_max": [round(float(np.median(td)) * 1e3, 3), round(float(td.max()) * 1e3, 3)],
            "cpu_port_encode_MiBps": round(len(raw) * k / ce / mib, 1), "cpu_port_decode_MiBps": round(len(raw) * j / cd / mib, 1),
            "readme_i5_2500k_decode_MiBps": pub[0] if pub else None, "readme_i5_2500k_encode_MiBps": pub[1] if pub else None,
        })
    print(json.dumps({"table": "snappy per file (BASELINE config 4)", "copies_per_batch": R, "protocol":
                      "each file alone as R independent streams resident in HBM, 2 warm-up + 20 samples (Criterion's count); wall time of "
                      "the batch call incl. host orchestration; *_MiBps and *_sd_pct from the mean, *_GBps from the MEDIAN, "
                      "*_ms_median_max = [median, slowest sample]; outputs bit-exact vs the CPU port; CPU port = oracle/, 1 thread",
                      "rows": rows}), flush=True)


if __name__ == "__main__":
    main()
