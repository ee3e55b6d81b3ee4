"""N > 1 path on CPU. Two gloo ranks run the SAME functions bench.py runs on N GPUs for the strong-scaling workload
(lzfse_rust_amd.sharding: chunk_bounds -> shard -> process_shard -> gather_reports -> merge_reports -> check_against);
only the codec object differs: here the oracle stands in for the GPU (no GPU in this container), in bench.py it is
GpuCodec over the product's batch API. Also: bench.py's own launcher really starts N workers from a plain shell."""
import os
import socket
import subprocess
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleCodec:
    """Stand-in with the interface of bench.GpuCodec."""

    def __init__(self):
        from oracle_py import Oracle
        self.o = Oracle()

    def encode_batch(self, raws):
        return [self.o.encode(r) for r in raws]

    def decode_batch(self, encs, raw_lens):
        return [self.o.decode(e, cap=n) for e, n in zip(encs, raw_lens)]


def _worker(rank, world, port, total_len, chunk):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lzfse_rust_amd import sharding
    from oracle_py import seq_masked
    codec = OracleCodec()
    data = seq_masked(5, 0x03030303, total_len)
    n_chunks = len(sharding.chunk_bounds(total_len, chunk))
    report = sharding.process_shard(data, chunk, rank, world, codec)
    assert sorted(report) == sharding.shard(n_chunks, rank, world)
    reports = sharding.gather_reports(report, world, dist)      # the only cross-rank traffic: metadata
    if rank == 0:
        merged = sharding.merge_reports(reports, n_chunks)
        reference = sharding.process_shard(data, chunk, 0, 1, codec)   # ONE encoder over all chunks
        assert sharding.check_against(merged, reference)
        assert sum(v[0] for v in merged.values()) == total_len
        # a wrong stream in any rank's report is caught
        c0 = next(iter(reports[1]))
        broken = [dict(reports[0]), dict(reports[1])]
        broken[1][c0] = (broken[1][c0][0], broken[1][c0][1], "0" * 64)
        with pytest.raises(AssertionError):
            sharding.check_against(sharding.merge_reports(broken, n_chunks), reference)
        # and a chunk nobody reported
        with pytest.raises(AssertionError):
            sharding.merge_reports([reports[0]], n_chunks)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chunk_sharding():
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 3 * (1 << 20) + 12345, 1 << 20), nprocs=world, join=True)


def test_shard_partition_properties():
    from lzfse_rust_amd import sharding
    for n in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in sharding.shard(n, r, world))
            assert seen == list(range(n))
            for r in range(world):
                assert all(sharding.owner(i, world) == r for i in sharding.shard(n, r, world))
    assert sharding.chunk_bounds(10, 4) == [(0, 4), (4, 4), (8, 2)]
    assert sharding.chunk_bounds(1 << 30) == [(o, 4 << 20) for o in range(0, 1 << 30, 4 << 20)]


def test_bench_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell must start two workers itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1) before anything touches a GPU. The workers are replaced by a stub here."""
    stub = tmp_path / "bench.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    # keep the real launcher and argument parsing, end every worker right after its rank set-up
    marker = "    import torch\n    import torch.distributed as dist\n"
    assert marker in src
    src = src.replace(marker, "    print('RANKENV', os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], "
                              "os.environ['MASTER_ADDR'], flush=True)\n    return\n" + marker, 1)
    src = src.replace("ROOT = os.path.dirname(os.path.abspath(__file__))", f"ROOT = {ROOT!r}")
    stub.write_text(src)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(stub), "--gpus", "2", "--workload", "chunks1g"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith("RANKENV"))
    assert lines == ["RANKENV 0 0 2 127.0.0.1", "RANKENV 1 1 2 127.0.0.1"]
    # under an external launcher with a mismatching world size the message says what to run
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    stub2 = tmp_path / "bench2.py"
    stub2.write_text(open(os.path.join(ROOT, "bench.py")).read().replace("    import torch\n    import torch.distributed as dist\n",
                                                                        "    torch = dist = None\n", 1))
    out = subprocess.run([sys.executable, str(stub2), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "torch.distributed.run" in out.stderr
