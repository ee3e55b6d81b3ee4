"""N > 1 path on CPU: two gloo ranks shard independent streams with no data-path collective and
reproduce the single-process result. The codec calls are stood in by the oracle here (no GPU in
this container); the sharding / gather logic is the code bench.py and users run on N GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_len, chunk, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    from lzfse_rust_amd import sharding
    from oracle_py import Oracle, seq_masked
    o = Oracle()
    data = seq_masked(5, 0x03030303, total_len)
    bounds = sharding.chunk_bounds(total_len, chunk)
    mine = sharding.shard(len(bounds), rank, world)
    lens = torch.zeros(len(bounds), dtype=torch.int64)
    for c in mine:
        off, ln = bounds[c]
        enc = o.encode(data[off:off + ln])
        open(os.path.join(out_dir, f"chunk{c}.lzfse"), "wb").write(enc)
        lens[c] = len(enc)
    # the only cross-rank traffic: result sizes (metadata), like bench.py's timing reduce
    dist.all_reduce(lens)
    dist.barrier()
    if rank == 0:
        assert (lens > 0).all()
        out = bytearray()
        for c in range(len(bounds)):
            enc = open(os.path.join(out_dir, f"chunk{c}.lzfse"), "rb").read()
            assert len(enc) == int(lens[c])
            out += o.decode(enc)
        assert bytes(out) == data
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chunk_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 3 * (1 << 20) + 12345, 1 << 20, str(tmp_path)), nprocs=world, join=True)


def test_shard_partition_properties():
    from lzfse_rust_amd import sharding
    for n in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in sharding.shard(n, r, world))
            assert seen == list(range(n))
            for r in range(world):
                assert all(sharding.owner(i, world) == r for i in sharding.shard(n, r, world))
    assert sharding.chunk_bounds(10, 4) == [(0, 4), (4, 4), (8, 2)]
