"""Drop-in for the reference's ``Evaluation.py`` (LFW / CFP-FP / AgeDB-30 style verification).

Same function names, argument order and return types as ``Mobilefacenet/Evaluation.py:18-99``.
``getFeature`` runs the embeddings and the flip-TTA combine on the HIP device and returns CPU
tensors like the reference.  The 10-fold protocol itself is host-side numpy in the reference and
stays host-side numpy here, with the same array expressions for centring and scoring; only the
20,001-step threshold sweep is replaced by an exact sorted-count formulation that yields the
same accuracies, hence bit-identical thresholds, about 1000x faster.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .face_model import MobileFaceNet, l2_norm, tta_combine  # noqa: F401


def getAccuracy(scores, flags, threshold, method):
    """Evaluation.py:18-25 -- strict inequalities on both sides."""
    if method == "l2_distance":
        p = np.sum(scores[flags == 1] < threshold)
        n = np.sum(scores[flags == -1] > threshold)
    elif method == "cos_distance":
        p = np.sum(scores[flags == 1] > threshold)
        n = np.sum(scores[flags == -1] < threshold)
    else:
        raise ValueError(f"unknown method {method!r}")
    return 1.0 * (p + n) / len(scores)


def getThreshold(scores, flags, thrNum, method):
    """Evaluation.py:27-34 -- mean of every threshold on the grid k*3/thrNum (|k| <= thrNum) that
    attains the maximum accuracy.

    The reference evaluates getAccuracy at each of the 2*thrNum+1 thresholds.  #(s < t) and #(s > t)
    over a sorted array are two binary searches, so all accuracies come from four searchsorted
    calls.  The comparisons are made in the type numpy gives ``scores < thresholds[i]`` in the reference's
    loop -- float64 under NumPy >= 2 promotion, the array's own type under NumPy 1.x value-based casting
    of the scalar -- so every count, and the result, is identical under either.
    """
    if method not in ("l2_distance", "cos_distance"):
        raise ValueError(f"unknown method {method!r}")
    scores = np.asarray(scores)
    flags = np.asarray(flags)
    thresholds = np.arange(-thrNum, thrNum + 1) * 3.0 / thrNum
    cmp = np.result_type(scores[:0], thresholds[0])       # an array against a float64 SCALAR, as in the reference's loop
    pos = np.sort(scores[flags == 1].astype(cmp))
    neg = np.sort(scores[flags == -1].astype(cmp))
    grid = thresholds.astype(cmp)
    pos_lt = np.searchsorted(pos, grid, side="left")
    pos_gt = len(pos) - np.searchsorted(pos, grid, side="right")
    neg_lt = np.searchsorted(neg, grid, side="left")
    neg_gt = len(neg) - np.searchsorted(neg, grid, side="right")
    if np.isnan(pos).any() or np.isnan(neg).any():
        # NaN compares false both ways; searchsorted would put it last and count it as "greater"
        pos_gt = pos_gt - int(np.isnan(pos).sum())
        neg_gt = neg_gt - int(np.isnan(neg).sum())
    correct = (pos_lt + neg_gt) if method == "l2_distance" else (pos_gt + neg_lt)
    accuracys = 1.0 * correct / len(scores)
    max_index = accuracys == np.max(accuracys)
    return np.mean(thresholds[max_index])


def getFeature(net, dataloader, device, flip=True, keep_on_device=False, group_crops=None):
    """Evaluation.py:36-66.  ``dataloader`` yields lists ``[L, flip(L), R, flip(R)]`` of f32
    [b,3,112,112]; returns CPU f32 tensors (featureLs, featureRs) in loader order.
    ``keep_on_device=True`` (not in the reference) leaves them on the HIP device for
    ``evaluation_10_fold``'s device path.

    ``group_crops`` (not in the reference): consecutive loader batches go through ONE forward as long as
    they hold at most that many crops together (the reference harness's batch size 50 is 200 crops per
    forward, a quarter of what fills the chip).  A row of the output does not depend on the batch it was
    computed in -- bit for bit, for a network that says so (``net.batch_invariant``, or from
    ``net.batch_invariant_from()`` crops on) -- so the features are the ones a forward per loader batch gives:
    a loader batch below that size goes through a forward of its own.  Default: 1024 for such a network, else
    0 = one forward per loader batch, as the reference does."""
    invariant = bool(getattr(net, "batch_invariant", False))
    min_group = 1 if invariant else (net.batch_invariant_from(device) if hasattr(net, "batch_invariant_from") else None)
    if group_crops is None:
        group_crops = 1024 if min_group is not None else 0
    if min_group is None:
        min_group = 1
    featureLs, featureRs = [], []
    pending, pending_crops = [], 0

    def flush():
        flat = [d for det in pending for d in det]
        with torch.no_grad():
            res = torch.split(net(torch.cat(flat, 0)), [d.size(0) for d in flat], 0)   # one launch sequence for all of them
            at = 0
            for det in pending:
                r = res[at:at + len(det)]
                at += len(det)
                if flip:
                    featureLs.append(tta_combine(r[0], r[1]))
                    featureRs.append(tta_combine(r[2], r[3]))
                else:
                    featureLs.append(r[0])
                    featureRs.append(r[2])
        pending.clear()

    for det in dataloader:
        det = [d.to(device, non_blocking=True) for d in det]
        n = sum(d.size(0) for d in det)
        alone = n < min_group                              # too small to share a forward without changing its bits
        if pending and (alone or pending_crops + n > group_crops):
            flush()
            pending_crops = 0
        pending.append(det)
        pending_crops += n
        if alone:
            flush()
            pending_crops = 0
    if pending:
        flush()
    if not featureLs:
        return None, None
    featureL, featureR = torch.cat(featureLs, 0), torch.cat(featureRs, 0)
    return (featureL, featureR) if keep_on_device else (featureL.cpu(), featureR.cpu())


def getFeature_u8(net, pair_batches, device, flip=True, keep_on_device=False, max_batch=1024):
    """``getFeature`` for RAW crops (not in the reference): ``pair_batches`` yields ``(L, R)`` uint8 [b,112,112,3] HWC BGR
    host tensors (``data_set.dataloader.u8_pair_batches``).  ToTensor + Normalize and the mirrored copy of the flip
    test-time augmentation (Evaluation.py:50-52) happen inside the first kernel, and the host -> device copy of one
    batch runs under the forward of the batch before it (``streaming.EmbeddingStream``): a quarter of the bytes of the
    float route over PCIe and no host-side transform."""
    from .streaming import EmbeddingStream
    stream = EmbeddingStream(net, max_batch=max_batch, depth=3, tta=flip, device=torch.device(device))
    sizes = []

    def crops():
        for left, right in pair_batches:
            if left.shape[0] != right.shape[0] or 2 * left.shape[0] > max_batch:
                raise ValueError(f"pair batch of {left.shape[0]} / {right.shape[0]} crops (at most {max_batch // 2} pairs)")
            sizes.append(left.shape[0])
            yield torch.cat([left, right], 0)

    feats = [f for f in stream.embed(crops())]
    if not feats:
        return None, None
    featureL = torch.cat([f[:n] for f, n in zip(feats, sizes)], 0)
    featureR = torch.cat([f[n:] for f, n in zip(feats, sizes)], 0)
    return (featureL, featureR) if keep_on_device else (featureL.cpu(), featureR.cpu())


def evaluation_10_fold(featureL, featureR, dataset, method="l2_distance"):
    """Evaluation.py:68-99.  ``dataset`` needs ``.folds`` (0..9) and ``.flags`` (+-1).

    Keeps the reference's arithmetic line for line where it decides results: the validation-fold
    mean is subtracted from the arrays carried into the next fold (:86-89), so the centring
    accumulates across folds exactly as it does there.
    """
    if isinstance(featureL, torch.Tensor) and featureL.is_cuda:
        if featureL.dim() == 2 and featureL.shape[1] in (128, 256, 512):
            return _evaluation_10_fold_device(featureL, featureR, dataset, method)
        # widths the device kernel does not take (embedding_size 1024): the reference's host arithmetic
        featureL, featureR = featureL.cpu(), featureR.cpu()
    ACCs = np.zeros(10)
    threshold = np.zeros(10)
    fold = np.array(dataset.folds).reshape(1, -1)
    flags = np.array(dataset.flags).reshape(1, -1)
    featureLs = featureL.numpy() if isinstance(featureL, torch.Tensor) else np.asarray(featureL)
    featureRs = featureR.numpy() if isinstance(featureR, torch.Tensor) else np.asarray(featureR)
    for i in range(10):
        valFold = fold != i
        testFold = fold == i
        flags = np.squeeze(flags)
        mu = np.mean(np.concatenate((featureLs[valFold[0], :], featureRs[valFold[0], :]), 0), 0)
        mu = np.expand_dims(mu, 0)
        featureLs = featureLs - mu
        featureRs = featureRs - mu
        featureLs = featureLs / np.expand_dims(np.sqrt(np.sum(np.power(featureLs, 2), 1)), 1)
        featureRs = featureRs / np.expand_dims(np.sqrt(np.sum(np.power(featureRs, 2), 1)), 1)
        if method == "l2_distance":
            scores = np.sum(np.power((featureLs - featureRs), 2), 1)
        elif method == "cos_distance":
            scores = np.sum(np.multiply(featureLs, featureRs), 1)
        else:
            raise ValueError(f"unknown method {method!r}")
        threshold[i] = getThreshold(scores[valFold[0]], flags[valFold[0]], 10000, method)
        ACCs[i] = getAccuracy(scores[testFold[0]], flags[testFold[0]], threshold[i], method)
    return ACCs, threshold


def _evaluation_10_fold_device(featureL, featureR, dataset, method):
    """evaluation_10_fold with the features resident on the HIP device: centring, normalisation and
    scoring of every fold run there (``ev_fold_scores``, float32 operations in numpy's order, so the
    scores equal the host path's bit for bit); the threshold search and the accuracy count take the
    P scores of a fold on the host."""
    if method not in ("l2_distance", "cos_distance"):
        raise ValueError(f"unknown method {method!r}")
    if not (isinstance(featureR, torch.Tensor) and featureR.device == featureL.device):
        raise RuntimeError("evaluation_10_fold: featureL and featureR must be on the same device")
    if featureL.shape != featureR.shape or featureL.dim() != 2:
        raise RuntimeError(f"evaluation_10_fold: feature shapes {tuple(featureL.shape)} / {tuple(featureR.shape)}")
    lib = _lib.load()
    dev = featureL.device
    P, D = int(featureL.shape[0]), int(featureL.shape[1])
    fold = np.array(dataset.folds).reshape(-1)
    flags = np.array(dataset.flags).reshape(-1)
    if fold.shape[0] != P or flags.shape[0] != P:
        raise RuntimeError(f"evaluation_10_fold: {P} pairs but {fold.shape[0]} folds / {flags.shape[0]} flags")
    ACCs = np.zeros(10)
    threshold = np.zeros(10)
    with torch.cuda.device(dev):
        # private copies: the reference rebinds its local arrays and leaves the caller's untouched
        fl = featureL.detach().to(torch.float32).contiguous().clone()
        fr = featureR.detach().to(torch.float32).contiguous().clone()
        mu = torch.empty(D, dtype=torch.float32, device=dev)
        scores = torch.empty(P, dtype=torch.float32, device=dev)
        for i in range(10):
            val = fold != i
            test = fold == i
            mask = torch.from_numpy(val.astype(np.uint8)).to(dev)
            _lib.check(lib.ev_fold_scores(fl.data_ptr(), fr.data_ptr(), mask.data_ptr(), int(val.sum()), P, D,
                                          0 if method == "l2_distance" else 1, mu.data_ptr(), scores.data_ptr(),
                                          _lib.current_stream_ptr(dev)), "ev_fold_scores")
            sc = scores.cpu().numpy()
            threshold[i] = getThreshold(sc[val], flags[val], 10000, method)
            ACCs[i] = getAccuracy(sc[test], flags[test], threshold[i], method)
    return ACCs, threshold
