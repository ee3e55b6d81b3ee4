"""Chunk sharding for multi-GPU runs (BASELINE.json config 5).

A large input is cut into fixed-size chunks, each encoded as its own LZFSE stream (tables are reset per call in the
reference, frontend_bytes.rs:113-119, so chunks are independent); chunk c goes to rank c mod world. No data-path
collective exists: ranks only exchange result metadata (sizes, hashes) at the end. The chunk framing is this build's own:
a plain LzfseDecoder decodes each chunk but not their concatenation (EOS rule, decoder.rs:93-95).

bench.py (one process per GPU) and tests/test_multirank.py (two gloo ranks on CPU, a CPU codec stood in) run the same
functions below; only the codec object differs.
"""
import hashlib

CHUNK_BYTES = 4 << 20


def chunk_bounds(total_len, chunk=CHUNK_BYTES):
    """[(offset, length)] of the independent streams a large input is cut into."""
    return [(o, min(chunk, total_len - o)) for o in range(0, total_len, chunk)] or [(0, 0)]


def shard(n_items, rank, world):
    """Indices of the items rank `rank` owns: item c -> rank c mod world (round robin)."""
    return list(range(rank, n_items, world))


def owner(item, world):
    return item % world


def process_shard(data, chunk, rank, world, codec):
    """Encode and decode this rank's chunks of `data` through `codec` (encode_batch(list of bytes) -> list of bytes,
    decode_batch(list of streams, list of raw lengths) -> list of bytes). Returns the rank's report:
    {chunk index: (raw length, stream length, sha256 of the stream)}; raises if a chunk does not round-trip."""
    bounds = chunk_bounds(len(data), chunk)
    mine = shard(len(bounds), rank, world)
    raws = [bytes(data[o:o + n]) for o, n in (bounds[c] for c in mine)]
    encs = codec.encode_batch(raws)
    decs = codec.decode_batch(encs, [len(r) for r in raws])
    report = {}
    for c, r, e, d in zip(mine, raws, encs, decs):
        if bytes(d) != r:
            raise AssertionError(f"chunk {c} does not round-trip on rank {rank}")
        report[c] = (len(r), len(e), hashlib.sha256(bytes(e)).hexdigest())
    return report


def merge_reports(reports, n_chunks):
    """Union of the ranks' reports (what rank 0 gathers); every chunk must be reported exactly once."""
    out = {}
    for rep in reports:
        for c, v in rep.items():
            if c in out:
                raise AssertionError(f"chunk {c} reported twice")
            out[int(c)] = tuple(v)
    missing = [c for c in range(n_chunks) if c not in out]
    if missing:
        raise AssertionError(f"chunks without an owner: {missing[:8]}")
    return out


def check_against(merged, reference):
    """Per-chunk equality of the gathered streams with a reference list made by ONE encoder (rank 0's; the CPU test
    injects its own): same lengths, same SHA-256."""
    bad = [c for c in sorted(reference) if merged.get(c) != tuple(reference[c])]
    if bad:
        raise AssertionError(f"{len(bad)} chunk streams differ from the reference list: {bad[:8]}")
    return True


def gather_reports(report, world, dist=None):
    """all_gather of the per-rank report dicts (metadata only) through torch.distributed when world > 1."""
    if world == 1 or dist is None:
        return [report]
    out = [None] * world
    dist.all_gather_object(out, report)
    return out
