"""Seeded synthetic weights, crops, galleries and pair sets.

The reference ships neither trained MobileFaceNet weights nor the LFW /
AgeDB / CFP-FP images, so every test and benchmark runs on data made here.
Everything is drawn from ``numpy.random.default_rng(seed)`` so that the same
seed gives the same arrays in any process and on any machine.

The weight recipe is chosen so that activations keep an O(1) scale through
all 50 layers, as they do in a trained network whose BatchNorm statistics
match its activations: conv weights are variance preserving for their fan-in
(with the PReLU attenuation compensated), while the BatchNorm scale, shift,
running mean and running variance and the PReLU slopes are all non-trivial so
that the load-time folding is exercised.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import layers as L


def make_state_dict(seed: int = 1, embedding_size: int = 512) -> Dict[str, np.ndarray]:
    """Random MobileFaceNet state_dict with the reference's 333 keys (numpy arrays)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for c in L.all_convs():
        fan_in = (c.cin // c.groups) * c.k * c.k
        # PReLU with slope a keeps (1+a^2)/2 of a zero-mean signal's power; a ~ U(.1,.4).
        gain = 1.85 if c.prelu else 1.0
        if c.prefix.endswith(".project"):
            gain = 0.6  # keeps the residual stream from growing block after block
        std = np.sqrt(gain / fan_in)
        sd[c.prefix + ".conv.weight"] = rng.normal(0.0, std, c.weight_shape).astype(np.float32)
        sd[c.prefix + ".bn.weight"] = rng.uniform(0.5, 1.5, c.cout).astype(np.float32)
        sd[c.prefix + ".bn.bias"] = rng.normal(0.0, 0.1, c.cout).astype(np.float32)
        sd[c.prefix + ".bn.running_mean"] = rng.normal(0.0, 0.1, c.cout).astype(np.float32)
        sd[c.prefix + ".bn.running_var"] = rng.uniform(0.5, 2.0, c.cout).astype(np.float32)
        sd[c.prefix + ".bn.num_batches_tracked"] = np.array(1000, dtype=np.int64)
        if c.prelu:
            sd[c.prefix + ".prelu.weight"] = rng.uniform(0.1, 0.4, c.cout).astype(np.float32)
    sd["linear.weight"] = rng.normal(0.0, 1.0 / np.sqrt(L.FEAT_C), (embedding_size, L.FEAT_C)).astype(np.float32)
    sd["bn.weight"] = rng.uniform(0.5, 1.5, embedding_size).astype(np.float32)
    sd["bn.bias"] = rng.normal(0.0, 0.1, embedding_size).astype(np.float32)
    sd["bn.running_mean"] = rng.normal(0.0, 0.1, embedding_size).astype(np.float32)
    sd["bn.running_var"] = rng.uniform(0.5, 2.0, embedding_size).astype(np.float32)
    sd["bn.num_batches_tracked"] = np.array(1000, dtype=np.int64)
    # order as the reference module registers them
    return {k: sd[k] for k, _, _ in L.state_dict_keys(embedding_size)}


def with_embedding_statistics(sd: Dict[str, np.ndarray], mean: np.ndarray, var: np.ndarray) -> Dict[str, np.ndarray]:
    """Copy of ``sd`` whose final BatchNorm1d (``bn``, face_model.py:92) carries the given running statistics.

    In a trained network those statistics ARE the mean and variance of the ``linear`` layer's output over the
    data, so embeddings are spread over the sphere.  With random weights and random statistics every input lands
    near one common direction; the verification protocol then subtracts that direction (Evaluation.py:84-89) and
    what is left is small, which magnifies storage rounding by an order of magnitude -- a property of untrained
    weights, not of the arithmetic under test.  Measure mean / var of the ``linear`` output on sample crops (any
    implementation of the network will do) and pass them here to get weights that behave like trained ones."""
    out = dict(sd)
    out["bn.running_mean"] = np.asarray(mean, dtype=np.float32).copy()
    out["bn.running_var"] = np.maximum(np.asarray(var, dtype=np.float32), 1e-6).copy()
    return out


def make_crops(batch: int, seed: int = 0, quantised: bool = True) -> np.ndarray:
    """f32 [batch,3,112,112] in [-1,1], as ToTensor + Normalize(0.5,0.5) would give.

    ``quantised`` snaps values to k/127.5 - 1 (k = 0..255), i.e. real uint8 pixels.
    """
    rng = np.random.default_rng(seed)
    if quantised:
        px = rng.integers(0, 256, (batch, L.INPUT_C, L.INPUT_HW, L.INPUT_HW), dtype=np.uint8)
        return ((px.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
    return rng.uniform(-1.0, 1.0, (batch, L.INPUT_C, L.INPUT_HW, L.INPUT_HW)).astype(np.float32)


def make_smooth_crops(batch: int, seed: int = 0) -> np.ndarray:
    """Crops with spatial structure (low-frequency blobs plus pixel noise)."""
    rng = np.random.default_rng(seed)
    coarse = rng.uniform(-1.0, 1.0, (batch, 3, 14, 14)).astype(np.float32)
    up = np.repeat(np.repeat(coarse, 8, axis=2), 8, axis=3)
    x = 0.8 * up + 0.2 * rng.uniform(-1.0, 1.0, up.shape).astype(np.float32)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def unit_rows(n: int, d: int = 512, seed: int = 2) -> np.ndarray:
    """n rows of N(0, I_d), L2-normalised, f32."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return g.astype(np.float32)


def planted_queries(gallery: np.ndarray, q: int, seed: int = 4, target_cos: float = 0.7
                    ) -> Tuple[np.ndarray, np.ndarray]:
    """Half the queries are a gallery row plus noise (true match known), half random.

    Returns (queries f32 [q,d] unit rows, truth int64 [q] with -1 for the random half).
    """
    rng = np.random.default_rng(seed)
    n, d = gallery.shape
    truth = np.full(q, -1, dtype=np.int64)
    out = np.empty((q, d), dtype=np.float32)
    half = q // 2
    picks = rng.integers(0, n, half)
    noise = rng.standard_normal((half, d), dtype=np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    s = np.float32(np.sqrt(1.0 / target_cos ** 2 - 1.0))
    out[:half] = gallery[picks] + s * noise
    truth[:half] = picks
    out[half:] = rng.standard_normal((q - half, d), dtype=np.float32)
    out /= np.linalg.norm(out, axis=1, keepdims=True)
    return out.astype(np.float32), truth


class PairSet:
    """Stand-in for the reference's pair datasets: only ``folds`` and ``flags`` are consumed
    by evaluation_10_fold (``Evaluation.py:73-74``).  Fold = i // (pairs/10), half same / half
    different inside every fold block, as in the LFW pair file (``dataloader.py:36-45``)."""

    def __init__(self, pairs: int):
        if pairs % 20:
            raise ValueError("pairs must be a multiple of 20")
        per = pairs // 10
        self.folds = [i // per for i in range(pairs)]
        self.flags = [1 if (i % per) < per // 2 else -1 for i in range(pairs)]

    def __len__(self) -> int:
        return len(self.folds)


def planted_features(pairs: int, seed: int = 5, same_cos: float = 0.55, diff_cos: float = 0.15,
                     jitter: float = 0.18, d: int = 512) -> Tuple[np.ndarray, np.ndarray, PairSet]:
    """Unit feature pairs with controlled same / different cosine so that the 10-fold
    accuracy lands in the 90-99 % band where a 0.1 pt tolerance is discriminating."""
    rng = np.random.default_rng(seed)
    ds = PairSet(pairs)
    flags = np.asarray(ds.flags)
    base = rng.standard_normal((pairs, d), dtype=np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    other = rng.standard_normal((pairs, d), dtype=np.float32)
    other -= np.sum(other * base, axis=1, keepdims=True) * base
    other /= np.linalg.norm(other, axis=1, keepdims=True)
    cos = np.where(flags == 1, same_cos, diff_cos) + jitter * rng.standard_normal(pairs)
    cos = np.clip(cos, -0.95, 0.95).astype(np.float32)[:, None]
    left = base
    right = cos * base + np.sqrt(1.0 - cos ** 2) * other
    # a shared offset makes the fold-wise mean-centring matter
    shift = rng.standard_normal((1, d), dtype=np.float32) * np.float32(0.02)
    left = left + shift
    right = right + shift
    left /= np.linalg.norm(left, axis=1, keepdims=True)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    return left.astype(np.float32), right.astype(np.float32), ds


def pair_crops(pairs: int, seed: int = 6, noise: float = 0.3) -> Tuple[np.ndarray, np.ndarray, PairSet]:
    """Synthetic verification images: same-pairs are two noisy copies of one template,
    different-pairs are two templates.  Returns (left, right) f32 [pairs,3,112,112]."""
    rng = np.random.default_rng(seed)
    ds = PairSet(pairs)
    flags = np.asarray(ds.flags)
    shape = (pairs, 3, L.INPUT_HW, L.INPUT_HW)
    coarse = rng.uniform(-1, 1, (pairs, 3, 14, 14)).astype(np.float32)
    t1 = np.repeat(np.repeat(coarse, 8, 2), 8, 3)
    coarse2 = rng.uniform(-1, 1, (pairs, 3, 14, 14)).astype(np.float32)
    t2 = np.repeat(np.repeat(coarse2, 8, 2), 8, 3)
    left = t1 + noise * rng.standard_normal(shape, dtype=np.float32)
    same = (flags == 1)[:, None, None, None]
    right = np.where(same, t1, t2) + noise * rng.standard_normal(shape, dtype=np.float32)
    return np.clip(left, -1, 1).astype(np.float32), np.clip(right, -1, 1).astype(np.float32), ds
