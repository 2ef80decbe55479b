"""The N > 1 path (contiguous shards, no data-path collective).

CPU tier  world_size-2 gloo: shard_range, the global-index seeding and bench.py's barrier / max-over-ranks / all-ranks-ok /
          per-rank gather plumbing; the
          per-shard compute is the oracle there (the HIP engine cannot run without a GPU), i.e. the checker stands in.
GPU tier  the same two-rank job with every shard computed by the HIP ENGINE (both ranks on device 0, gloo for the barrier,
          as bench.py's rehearsal mode does) against the single-context bytes and the oracle."""
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


def _shard_inputs(bench, lo, hi):
    return tuple(np.frombuffer(b"".join(bench.expand(lbl, i) for i in range(lo, hi)), np.uint8).reshape(-1, 32).copy()
                 for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))


def _digests(c, K):
    return np.frombuffer(b"".join(hashlib.sha256(bytes(c[i]) + bytes(K[i])).digest() for i in range(len(K))), np.uint8)


def _worker(rank, world, port, n, out_dir, use_engine):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    import bench
    pkg = ge.load_package()
    if use_engine:
        # exactly bench.py's set-up: fewer GPUs than ranks -> ranks share device 0, gloo
        r2, w2, local = bench.dist_setup(world, rehearse=True)
        assert (r2, w2, local) == (rank, world, 0) and bench.SHARED_GPU
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pkg.shard_range(n, rank, world)
    d, z, m = _shard_inputs(bench, lo, hi)
    if use_engine:
        eng = pkg.MLKEM(512, device=0, chunk_items=64)     # the HIP engine; fails loudly if the extension is missing
        ek, dk = eng.keygen(torch.from_numpy(d), torch.from_numpy(z))
        c, K = eng.encaps(ek, torch.from_numpy(m))
        K2, st = eng.decaps(dk, c)
        torch.cuda.synchronize()
        c, K, K2, st = (t.cpu().numpy() for t in (c, K, K2, st))
        eng.close()
    else:
        from oracle.loader import Oracle
        orc = Oracle()
        ek, dk = orc.keygen(512, d, z)
        c, K = orc.encaps(512, ek, m)
        K2, st = orc.decaps(512, dk, c)
    assert (K2 == K).all() and (st == 0).all()
    bench.barrier(world)
    t = bench.max_over_ranks(float(rank + 1), world, torch.device("cuda", 0) if use_engine else torch.device("cpu"))
    assert t == float(world)
    # bench.py's gate is the AND over ALL ranks, and every rank learns it (a wrong byte on rank 1 must fail the job)
    dv = torch.device("cuda", 0) if use_engine else torch.device("cpu")
    assert bench.all_ranks_ok(True, dv) is True
    assert bench.all_ranks_ok(rank != world - 1, dv) is False
    # the in-job N = 1 anchor: every rank runs SOLO_STEPS steps alone, in rank order, the others waiting at a barrier;
    # the step appends to a file shared by the ranks, so an overlap of two ranks' solo phases would interleave their marks
    log = os.path.join(out_dir, "solo.log")
    def solo_step():
        with open(log, "a") as f:
            f.write("%d\n" % rank)
    solo = bench.solo_anchor(solo_step, lambda: None, rank, world, 1000)
    assert solo is not None and solo > 0
    bench.barrier(world)
    marks = open(log).read().split()
    assert marks == [str(r) for r in range(world) for _ in range(bench.SOLO_STEPS)], marks
    per = bench.gather_per_gpu({"rank": rank, "value": 10.0 * (rank + 1), "solo_value": 20.0 * (rank + 1)})
    assert [p["rank"] for p in per] == list(range(world)) and per[-1]["value"] == 10.0 * world
    if rank == 0:
        anc = bench.scaling_anchor(sum(p["value"] for p in per), per)
        assert abs(anc["efficiency"] - 0.5) < 1e-12 and anc["per_gpu_efficiency"] == [0.5] * world and anc["solo_steps"] == bench.SOLO_STEPS
        assert bench.scaling_anchor(1.0, [{"value": 1.0}]) is None and bench.solo_anchor(solo_step, lambda: None, 0, 1, 1) is None
    # gather per-item digests on rank 0 (test-only collective; the product's data path has none)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, _digests(c, K).tobytes()))
    if rank == 0:
        np.save(os.path.join(out_dir, "digests.npy"), np.frombuffer(b"".join(g[2] for g in sorted(gathered)), np.uint8))
        spans = sorted((g[0], g[1]) for g in gathered)
        assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    dist.destroy_process_group()


def _oracle_digests(oracle, n):
    import bench
    d, z, m = _shard_inputs(bench, 0, n)
    ek, dk = oracle.keygen(512, d, z)
    c, K = oracle.encaps(512, ek, m)
    return _digests(c, K)


def test_two_rank_shards_equal_single_process(tmp_path, oracle):
    n, world = 9, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path), False), nprocs=world, join=True)
    assert (np.load(tmp_path / "digests.npy") == _oracle_digests(oracle, n)).all()


@pytest.mark.gpu
def test_two_rank_shards_computed_by_the_hip_engine(tmp_path, oracle):
    """Two freshly spawned ranks, each with its own engine context on device 0, each computing shard_range(n, rank, 2) of
    the globally indexed batch: concatenated, the shards equal the oracle's bytes for the whole batch and the
    single-context engine's."""
    n, world = 777, 2     # ragged: 389 + 388, several 64-item chunks per rank
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path), True), nprocs=world, join=True)   # before this process touches the GPU
    got = np.load(tmp_path / "digests.npy")
    assert (got == _oracle_digests(oracle, n)).all()
    import bench
    pkg = ge.load_package()
    d, z, m = _shard_inputs(bench, 0, n)
    eng = pkg.MLKEM(512, device=0)
    ek, dk = eng.keygen(torch.from_numpy(d), torch.from_numpy(z))
    c, K = eng.encaps(ek, torch.from_numpy(m))
    assert (got == _digests(c.cpu().numpy(), K.cpu().numpy())).all()
    eng.close()


def test_bench_requires_torchrun_for_multi_gpu(monkeypatch):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.dist_setup(2)
