"""CPU oracle for facebank matching and top-k search.  TEST INFRASTRUCTURE ONLY.

Restates the ten lines every demo of the reference repeats inline
(``Mobilefacenet/cam_demo.py:96-105``, ``Video_demo.py:99-108``,
``MTCNN_MobileFaceNet.py:76-85``) and a brute-force top-k used to check the
MFMA search.  Nothing under ``face_recognition_amd/`` may import this file.

Pinning: the expressions in ``match_broadcast`` are the reference's own
expressions, and ``tests/golden/facebank_match.npz`` holds their result on the
``facebank.pth`` the reference ships (5 identities), produced by
``tests/golden/make_golden.py``.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch


def match_broadcast(source_embs: torch.Tensor, targets: torch.Tensor, threshold: float = 60.0
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """cam_demo.py:97-100,105.  Returns (minimum f32[Q], min_idx i64[Q] with -1 = unknown, score_100 f32[Q])."""
    diff = source_embs.unsqueeze(-1) - targets.transpose(1, 0).unsqueeze(0)   # :97  [Q,D,N]
    dist = torch.sum(torch.pow(diff, 2), dim=1)                               # :98  [Q,N]
    minimum, min_idx = torch.min(dist, dim=1)                                 # :99
    min_idx = min_idx.clone()
    min_idx[minimum > ((threshold - 156) / (-80))] = -1                       # :100
    score_100 = torch.clamp(minimum * -80 + 156, 0, 100)                      # :105
    return minimum, min_idx, score_100


def l2_topk_bruteforce(queries: np.ndarray, gallery: np.ndarray, k: int = 5, block: int = 4096
                       ) -> Tuple[np.ndarray, np.ndarray]:
    """Exact squared-L2 top-k in float64: (dist f64[Q,k] ascending, idx i64[Q,k]); ties -> lowest index.

    The gallery is converted to float64 one block at a time and each block contributes only its k
    smallest candidates per query (all candidates of a row whose k-th value is tied), so a 10^6-row
    gallery costs seconds and a block's worth of memory."""
    q = np.asarray(queries, dtype=np.float64)
    qn = np.sum(q * q, axis=1)[:, None]
    nq = q.shape[0]
    best_d = np.full((nq, 0), np.inf)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    for s in range(0, gallery.shape[0], block):
        gb = np.asarray(gallery[s:s + block], dtype=np.float64)
        d = qn + np.sum(gb * gb, axis=1)[None, :] - 2.0 * (q @ gb.T)
        nb = gb.shape[0]
        if nb > 4 * k:
            part = np.argpartition(d, k - 1, axis=1)[:, :k]
            kth = np.take_along_axis(d, part, axis=1).max(axis=1)
            tied = np.nonzero((d <= kth[:, None]).sum(axis=1) > k)[0]   # rows whose k-th value repeats
            cand_d = np.take_along_axis(d, part, axis=1)
            cand_i = part.astype(np.int64) + s
            for r in tied:                                             # keep the lowest indices of the tie
                o = np.lexsort((np.arange(nb), d[r]))[:k]
                cand_d[r], cand_i[r] = d[r, o], o + s
        else:
            cand_d = d
            cand_i = np.broadcast_to(np.arange(s, s + nb, dtype=np.int64)[None, :], d.shape)
        cd = np.concatenate([best_d, cand_d], axis=1)
        ci = np.concatenate([best_i, cand_i], axis=1)
        order = np.lexsort((ci, cd), axis=1)[:, :k]          # primary key distance, then index
        best_d = np.take_along_axis(cd, order, axis=1)
        best_i = np.take_along_axis(ci, order, axis=1)
    return best_d, best_i


def merge_topk(dist_parts: np.ndarray, idx_parts: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge R per-shard lists [R,Q,k'] into one [Q,k]; ascending distance, ties -> lowest index.
    Entries with index < 0 are empty slots and sort last."""
    r, q, kk = dist_parts.shape
    d = np.transpose(dist_parts, (1, 0, 2)).reshape(q, r * kk).astype(np.float64)
    i = np.transpose(idx_parts, (1, 0, 2)).reshape(q, r * kk).astype(np.int64)
    d = np.where(i < 0, np.inf, d)
    ikey = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.lexsort((ikey, d), axis=1)[:, :k]
    return np.take_along_axis(d, order, axis=1), np.take_along_axis(i, order, axis=1)
