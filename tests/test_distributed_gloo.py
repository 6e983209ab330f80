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
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), d=d.numpy(), i=i.numpy(), allemb=allemb.numpy())
    finally:
        dist.destroy_process_group()


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
    assert r0["i"][0, 0] == 10 and r0["i"][0, 1] == 4000
    assert np.allclose(r0["d"], bd, atol=1e-5)
    assert np.array_equal(r0["allemb"][:, 0], np.arange(11, dtype=np.float32))
