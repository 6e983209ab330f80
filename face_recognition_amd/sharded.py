"""Multi-GPU layout: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

Only two things shard, and they shard differently:

* embedding is data parallel -- every rank embeds its own contiguous slice of the crops with a
  full copy of the (2.4 MB) weights; there is NO collective on that path;
* 1:N search is gallery-row parallel -- rank r owns rows [r*N/R, (r+1)*N/R) (``index_base`` =
  first row), the queries are replicated, every rank computes its local top-k with global
  indices, and ONE all-gather of (distance, index) pairs (Q*k*8 bytes per rank) followed by a
  local merge gives every rank the global top-k.  Ordering (ascending distance, ties to the
  lowest index) is the same in the shard search and in the merge, so the result equals the
  single-GPU search of the whole gallery.

The search and merge steps are injectable so that the partition / gather / merge plumbing can be
exercised with the gloo backend on hosts without a GPU; by default they are the HIP kernels.
"""
from __future__ import annotations

import hashlib
import time
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of ``total`` items: first (total % world) ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0,{world})")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def result_digest(idx: torch.Tensor) -> str:
    """Digest of a search result's index tensor [Q,k] (row-major int32 bytes, first 16 hex digits of their SHA-256).
    The sharded search returns the same indices for any number of shards (ties go to the lowest global index in the
    shard search and in the merge alike), so runs at different world sizes must print the same digest."""
    a = idx.detach().to("cpu", torch.int32).contiguous().numpy()
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def _hip_merge(dist_parts: torch.Tensor, idx_parts: torch.Tensor, k: int):
    from .facebank import merge_topk
    return merge_topk(dist_parts, idx_parts, k)


def allgather_topk(local_dist: torch.Tensor, local_idx: torch.Tensor, group=None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather per-shard candidate lists: ([Q,k] f32, [Q,k] i32) -> ([R,Q,k] f32, [R,Q,k] i32).

    Distances and indices travel in one message: the f32 bits are viewed as int32 and packed next
    to the indices, so each rank contributes a single [Q,2k] int32 buffer."""
    world = dist.get_world_size(group)
    q, k = local_dist.shape
    packed = torch.empty((q, 2 * k), dtype=torch.int32, device=local_dist.device)
    packed[:, :k] = local_dist.contiguous().view(torch.int32)
    packed[:, k:] = local_idx.to(torch.int32)
    if packed.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves device tensors through the host itself.  Two gloo ranks that share ONE GPU (the one-GPU rehearsal
        # of the multi-rank path) block for ever inside its device path when the library's launches are still queued
        # and nothing torch launched follows them; the copy to the host synchronises first, which avoids that
        host = torch.empty((world, q, 2 * k), dtype=torch.int32)
        dist.all_gather_into_tensor(host.view(world * q, 2 * k), packed.cpu(), group=group)
        gathered = host.to(packed.device)
    else:
        gathered = torch.empty((world, q, 2 * k), dtype=torch.int32, device=local_dist.device)
        dist.all_gather_into_tensor(gathered.view(world * q, 2 * k), packed, group=group)
    d = gathered[:, :, :k].contiguous().view(torch.float32)
    i = gathered[:, :, k:].contiguous()
    return d, i


class ShardedGallery:
    """This rank's shard of a row-sharded gallery plus the gather / merge step.

    ``local_search(queries, k) -> (dist f32 [Q,k], idx int32 [Q,k])`` must report GLOBAL row indices
    (the HIP ``Gallery`` does when built with ``index_base`` = first row of the shard).
    """

    def __init__(self, total_rows: int, local_search: Callable, group=None,
                 merge: Optional[Callable] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_rows = int(total_rows)
        self.row_begin, self.row_end = shard_bounds(self.total_rows, self.world, self.rank)
        self.local_search = local_search
        self.merge = merge or _hip_merge

    @classmethod
    def from_rows(cls, local_rows: torch.Tensor, total_rows: int, store: str = "f16", group=None):
        """Build the HIP shard from this rank's rows (rows [row_begin,row_end) of the whole gallery)."""
        from .facebank import Gallery
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        b, e = shard_bounds(total_rows, world, rank)
        if local_rows.shape[0] != e - b:
            raise RuntimeError(f"rank {rank} owns rows [{b},{e}) but was given {local_rows.shape[0]} rows")
        g = Gallery(local_rows, store=store, index_base=b)
        sg = cls(total_rows, g.search, group)
        sg.gallery = g
        return sg

    def search(self, queries: torch.Tensor, k: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
        """Global top-k for replicated ``queries``: (dist f32 [Q,k], idx int32 [Q,k]); same on every rank."""
        ld, li = self.local_search(queries, k)
        if self.world == 1:
            return ld, li
        gd, gi = allgather_topk(ld, li, self.group)
        return self.merge(gd, gi, k)


    def search_timed(self, queries: torch.Tensor, k: int = 5) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, float]]:
        """``search`` with the three steps timed apart (host clock, a device synchronisation after each: a measurement
        aid, slower than ``search``): milliseconds of this rank's shard search, of the all-gather and of the merge."""
        def sync():
            if queries.is_cuda:
                torch.cuda.synchronize(queries.device)
        sync()
        t0 = time.perf_counter()
        ld, li = self.local_search(queries, k)
        sync()
        t1 = time.perf_counter()
        if self.world == 1:
            return ld, li, {"local_search_ms": (t1 - t0) * 1e3, "allgather_ms": 0.0, "merge_ms": 0.0}
        gd, gi = allgather_topk(ld, li, self.group)
        sync()
        t2 = time.perf_counter()
        d, i = self.merge(gd, gi, k)
        sync()
        t3 = time.perf_counter()
        return d, i, {"local_search_ms": (t1 - t0) * 1e3, "allgather_ms": (t2 - t1) * 1e3, "merge_ms": (t3 - t2) * 1e3}


def split_batch(total: int, group=None) -> Tuple[int, int]:
    """Slice of a batch of ``total`` crops this rank embeds (data parallel, no collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    return shard_bounds(total, world, rank)


def gather_embeddings(local_emb: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Optional: assemble [total,E] embeddings on every rank from the per-rank slices of ``split_batch``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_emb
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    maxn = max(e - b for b, e in sizes)
    pad = torch.zeros((maxn, local_emb.shape[1]), dtype=local_emb.dtype, device=local_emb.device)
    pad[: local_emb.shape[0]] = local_emb
    if pad.is_cuda and dist.get_backend(group) == "gloo":     # as in allgather_topk
        host = torch.empty((world * maxn, local_emb.shape[1]), dtype=local_emb.dtype)
        dist.all_gather_into_tensor(host, pad.cpu(), group=group)
        out = host.to(local_emb.device)
    else:
        out = torch.empty((world * maxn, local_emb.shape[1]), dtype=local_emb.dtype, device=local_emb.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * maxn: r * maxn + (e - b)] for r, (b, e) in enumerate(sizes)], 0)


def get_feature_sharded(net, batches_for: Callable, total_pairs: int, device, flip: bool = True, group=None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Verification features with the PAIRS split over the ranks (Evaluation.py:36-66 run data parallel): this rank embeds
    pairs ``split_batch(total_pairs)`` -- ``batches_for(lo, hi)`` yields their loader batches ``[L, flip(L), R,
    flip(R)]`` -- and one all-gather per side assembles ``featureLs`` / ``featureRs`` [total_pairs, E] in pair order on
    every rank, on the host, where the fold-by-fold threshold search of ``evaluation_10_fold`` runs.  No other
    collective; the result does not depend on the number of ranks."""
    from .Evaluation import getFeature
    lo, hi = split_batch(total_pairs, group)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    fl = fr = None
    if hi > lo:
        fl, fr = getFeature(net, batches_for(lo, hi), device, flip=flip, keep_on_device=True)
    if world > 1:
        # a rank without pairs (fewer pairs than ranks) still takes part in the gathers: it needs the feature width
        on = torch.device(device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        width = torch.tensor([0 if fl is None else fl.shape[1]], dtype=torch.int64, device=on)
        dist.all_reduce(width, op=dist.ReduceOp.MAX, group=group)
        if fl is None:
            fl = fr = torch.zeros((0, int(width.item())), dtype=torch.float32, device=torch.device(device))
    if fl is None:
        raise ValueError("get_feature_sharded: no pairs")
    return gather_embeddings(fl, total_pairs, group).cpu(), gather_embeddings(fr, total_pairs, group).cpu()
