"""Stream sharding for multi-GPU runs: independent LZFSE streams (or fixed-size chunks of one large
input, each encoded as its own stream) are dealt to ranks; no data-path collective exists
(SURVEY.md 8e). Chunk framing is this build's own: a plain LzfseDecoder decodes each chunk but not
their concatenation (decoder.rs:93-95)."""

CHUNK_BYTES = 4 << 20


def chunk_bounds(total_len, chunk=CHUNK_BYTES):
    """[(offset, length)] of the independent streams a large input is cut into."""
    return [(o, min(chunk, total_len - o)) for o in range(0, total_len, chunk)] or [(0, 0)]


def shard(n_items, rank, world):
    """Indices of the items rank `rank` owns: item c -> rank c mod world (round robin)."""
    return list(range(rank, n_items, world))


def owner(item, world):
    return item % world
