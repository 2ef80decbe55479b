"""CPU tier: the N>1 path (contiguous shards, no data-path collective) with world_size-2 gloo.
Each rank takes shard_range(n, rank, world) of a globally indexed batch; because the GPU engine cannot run
here, the per-shard compute is the oracle (checker stand-in) — what is under test is the sharding, the
global-index seeding and bench.py's barrier / max-over-ranks plumbing."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle.loader import Oracle
    pkg = ge.load_package()
    lo, hi = pkg.shard_range(n, rank, world)
    d = np.frombuffer(b"".join(bench.expand("mlkem-bench-d", i) for i in range(lo, hi)), np.uint8).reshape(-1, 32)
    z = np.frombuffer(b"".join(bench.expand("mlkem-bench-z", i) for i in range(lo, hi)), np.uint8).reshape(-1, 32)
    m = np.frombuffer(b"".join(bench.expand("mlkem-bench-m", i) for i in range(lo, hi)), np.uint8).reshape(-1, 32)
    orc = Oracle()
    ek, dk = orc.keygen(512, d, z)
    c, K = orc.encaps(512, ek, m)
    K2, st = orc.decaps(512, dk, c)
    assert (K2 == K).all() and (st == 0).all()
    bench.barrier(world)
    t = bench.max_over_ranks(float(rank + 1), world, torch.device("cpu"))
    assert t == float(world)
    # gather per-item digests on rank 0 (test-only collective; the product's data path has none)
    dig = np.frombuffer(b"".join(hashlib.sha256(bytes(c[i]) + bytes(K[i])).digest() for i in range(hi - lo)), np.uint8)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, dig.tobytes()))
    if rank == 0:
        np.save(os.path.join(out_dir, "digests.npy"), np.frombuffer(b"".join(g[2] for g in sorted(gathered)), np.uint8))
        spans = sorted((g[0], g[1]) for g in gathered)
        assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path, oracle):
    import bench
    n, world = 9, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "digests.npy")
    d = np.frombuffer(b"".join(bench.expand("mlkem-bench-d", i) for i in range(n)), np.uint8)
    z = np.frombuffer(b"".join(bench.expand("mlkem-bench-z", i) for i in range(n)), np.uint8)
    m = np.frombuffer(b"".join(bench.expand("mlkem-bench-m", i) for i in range(n)), np.uint8)
    ek, dk = oracle.keygen(512, d, z)
    c, K = oracle.encaps(512, ek, m)
    want = np.frombuffer(b"".join(hashlib.sha256(bytes(c[i]) + bytes(K[i])).digest() for i in range(n)), np.uint8)
    assert (got == want).all()


def test_bench_requires_torchrun_for_multi_gpu(monkeypatch):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.dist_setup(2)
