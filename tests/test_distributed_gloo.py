"""world_size-2 run of the sharded search plumbing on CPU with the gloo backend.

The GPU kernels cannot run here, so the two injectable steps (per-shard search, merge) are the CPU
oracle's; what is under test is the partitioning, the index_base convention, the packed all-gather
and that the merged result equals a single-process brute-force search of the whole gallery."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from face_recognition_amd import sharded, synthetic
from oracle import matching_ref

N, Q, K = 5001, 37, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synthetic.unit_rows(N, 64, seed=3)
        g[4000] = g[10]                              # a tie across shards: lower index must win
        q = np.concatenate([g[[10]], synthetic.unit_rows(Q - 1, 64, seed=5)])
        b, e = sharded.shard_bounds(N, world, rank)

        def local_search(queries, k):
            d, i = matching_ref.l2_topk_bruteforce(queries.numpy(), g[b:e], k)
            return torch.from_numpy(d.astype(np.float32)), torch.from_numpy((i + b).astype(np.int32))

        def merge(dp, ip, k):
            d, i = matching_ref.merge_topk(dp.numpy(), ip.numpy(), k)
            return torch.from_numpy(d.astype(np.float32)), torch.from_numpy(i.astype(np.int32))

        sg = sharded.ShardedGallery(N, local_search, merge=merge)
        assert (sg.row_begin, sg.row_end) == (b, e)
        d, i = sg.search(torch.from_numpy(q), K)
        lo, hi = sharded.split_batch(11)
        emb = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 4)
        allemb = sharded.gather_embeddings(emb, 11)
        d2, i2, parts = sg.search_timed(torch.from_numpy(q), K)
        assert torch.equal(i2, i) and set(parts) == {"local_search_ms", "allgather_ms", "merge_ms"}
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), d=d.numpy(), i=i.numpy(), allemb=allemb.numpy(),
                 digest=np.array(sharded.result_digest(i)))
    finally:
        dist.destroy_process_group()


def _worker_small(rank, world, port, out_dir, n):
    """Uneven shards, some holding fewer than K rows: their lists carry (inf, -1) in the missing slots, as the HIP
    search reports them, and the merge must skip those."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synthetic.unit_rows(n, 64, seed=13)
        q = synthetic.unit_rows(9, 64, seed=15)
        b, e = sharded.shard_bounds(n, world, rank)

        def local_search(queries, k):
            d, i = matching_ref.l2_topk_bruteforce(queries.numpy(), g[b:e], k)
            dd = np.full((queries.shape[0], k), np.inf, dtype=np.float32)
            ii = np.full((queries.shape[0], k), -1, dtype=np.int32)
            dd[:, : d.shape[1]] = d
            ii[:, : i.shape[1]] = i + b
            return torch.from_numpy(dd), torch.from_numpy(ii)

        def merge(dp, ip, k):
            d, i = matching_ref.merge_topk(dp.numpy(), ip.numpy(), k)
            return torch.from_numpy(d.astype(np.float32)), torch.from_numpy(i.astype(np.int32))

        sg = sharded.ShardedGallery(n, local_search, merge=merge)
        d, i = sg.search(torch.from_numpy(q), K)
        # verification features with the pairs split over the ranks (3 pairs over 4 ranks: one rank has none)
        w = torch.randn((3 * 112 * 112, 8), generator=torch.Generator().manual_seed(1)) * 0.01
        net = lambda x: torch.nn.functional.normalize(x.reshape(x.shape[0], -1) @ w)      # noqa: E731
        for pairs, name in ((23, "v"), (3, "w")):
            left, right, _ = synthetic.pair_crops(40, seed=8)
            L, R = torch.from_numpy(left[:pairs]), torch.from_numpy(right[:pairs])

            def batches_for(lo, hi, L=L, R=R):
                for s0 in range(lo, hi, 4):
                    l, r = L[s0:min(s0 + 4, hi)], R[s0:min(s0 + 4, hi)]
                    yield [l, torch.flip(l, dims=[3]), r, torch.flip(r, dims=[3])]
            fl, fr = sharded.get_feature_sharded(net, batches_for, pairs, torch.device("cpu"), flip=True)
            np.savez(os.path.join(out_dir, f"{name}{rank}.npz"), fl=fl.numpy(), fr=fr.numpy())
        lo, hi = sharded.split_batch(10)              # 10 crops over 4 ranks: 3, 3, 2, 2
        emb = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 4)
        allemb = sharded.gather_embeddings(emb, 10)
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), d=d.numpy(), i=i.numpy(), allemb=allemb.numpy(), bounds=np.array([b, e]))
    finally:
        dist.destroy_process_group()


def test_sharded_search_four_ranks_uneven_and_short_shards(tmp_path):
    """World 4.  11 rows -> shards of 3, 3, 3, 2 rows, every one shorter than k = 5; 5003 rows -> 1251, 1251, 1251, 1250."""
    world = 4
    assert [sharded.shard_bounds(11, world, r) for r in range(world)] == [(0, 3), (3, 6), (6, 9), (9, 11)]
    assert [e - b for b, e in (sharded.shard_bounds(5003, world, r) for r in range(world))] == [1251, 1251, 1251, 1250]
    for n in (11, 5003):
        out = tmp_path / f"n{n}"
        out.mkdir()
        mp.spawn(_worker_small, args=(world, _free_port(), str(out), n), nprocs=world, join=True)
        g = synthetic.unit_rows(n, 64, seed=13)
        q = synthetic.unit_rows(9, 64, seed=15)
        bd, bi = matching_ref.l2_topk_bruteforce(q, g, K)
        rs = [np.load(out / f"s{r}.npz") for r in range(world)]
        for r in rs:
            assert np.array_equal(r["i"], rs[0]["i"]) and np.array_equal(r["d"], rs[0]["d"])
            assert np.array_equal(r["allemb"][:, 0], np.arange(10, dtype=np.float32))
        assert np.array_equal(rs[0]["i"], bi) and (rs[0]["i"] >= 0).all()
        assert np.allclose(rs[0]["d"], bd, atol=1e-5)
        assert sorted(tuple(r["bounds"]) for r in rs)[-1][1] == n
        # verification features: every rank holds all pairs in order, equal to one process embedding them all
        from face_recognition_amd import Evaluation
        w = torch.randn((3 * 112 * 112, 8), generator=torch.Generator().manual_seed(1)) * 0.01
        net = lambda x: torch.nn.functional.normalize(x.reshape(x.shape[0], -1) @ w)      # noqa: E731
        for pairs, name in ((23, "v"), (3, "w")):
            left, right, _ = synthetic.pair_crops(40, seed=8)
            L, R = torch.from_numpy(left[:pairs]), torch.from_numpy(right[:pairs])
            one = Evaluation.getFeature(net, [[L, torch.flip(L, dims=[3]), R, torch.flip(R, dims=[3])]], torch.device("cpu"), flip=True)
            for r in range(world):
                v = np.load(out / f"{name}{r}.npz")
                assert v["fl"].shape == (pairs, 8)
                assert np.allclose(v["fl"], one[0].numpy(), atol=1e-6) and np.allclose(v["fr"], one[1].numpy(), atol=1e-6)


def test_sharded_search_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = synthetic.unit_rows(N, 64, seed=3)
    g[4000] = g[10]
    q = np.concatenate([g[[10]], synthetic.unit_rows(Q - 1, 64, seed=5)])
    bd, bi = matching_ref.l2_topk_bruteforce(q, g, K)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["i"], r1["i"]) and np.array_equal(r0["d"], r1["d"])      # same on every rank
    assert np.array_equal(r0["i"], bi)
    # what bench.py prints as search.result_digest: the same string from every rank and from one process on the whole gallery
    one = sharded.ShardedGallery(N, lambda qq, k: tuple(torch.from_numpy(a) for a in (
        lambda d_, i_: (d_.astype(np.float32), i_.astype(np.int32)))(*matching_ref.l2_topk_bruteforce(qq.numpy(), g, k))))
    assert one.world == 1
    d1, i1 = one.search(torch.from_numpy(q), K)
    assert str(r0["digest"]) == str(r1["digest"]) == sharded.result_digest(i1)
    assert sharded.result_digest(i1) != sharded.result_digest(i1.flip(0))
    assert r0["i"][0, 0] == 10 and r0["i"][0, 1] == 4000
    assert np.allclose(r0["d"], bd, atol=1e-5)
    assert np.array_equal(r0["allemb"][:, 0], np.arange(11, dtype=np.float32))
