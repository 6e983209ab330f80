#!/usr/bin/env python3
"""Counterpart of the reference's ``Mobilefacenet/test_lfw_agedb_cfpfp.py``: load a checkpoint, embed
the LFW / CFP-FP / AgeDB-30 pair sets with flip-TTA at batch size 50 and print the 10-fold
``cos_distance`` accuracy and threshold of each (``test_lfw_agedb_cfpfp.py:57-78``).

The reference reads aligned jpgs and pair lists from disk (``:27-41``).  Given such data (``--lfw ROOT PAIRS``,
``--cfp-fp ROOT PAIRS``, ``--agedb30 ROOT PAIRS``) this harness reads it the same way (``data_set/dataloader.py``, a
``DataLoader`` with two workers).  Neither ships with the reference, so without those options
the pair sets are synthetic with the same shape: 6000 / 7000 / 6000 pairs, fold = i // (P/10)
(``data_set/dataloader.py:41,90,135``), half same / half different pairs per fold block, items
``[L, flip(L), R, flip(R)]`` of f32 [3,112,112] in [-1,1] (``dataloader.py:57-65``).  Same-pairs are
two noisy copies of one template, different-pairs two templates.

    python -m face_recognition_amd.test_lfw_agedb_cfpfp --ckpt Iter_555000_model.ckpt
    python -m face_recognition_amd.test_lfw_agedb_cfpfp --seed 1            # seeded random weights

Everything numeric runs in the HIP library (``getFeature`` -> MobileFaceNet forward,
``evaluation_10_fold`` on the device); there is no CPU path.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np
import torch

from . import synthetic
from .Evaluation import evaluation_10_fold, getFeature
from .face_model import MobileFaceNet

PHASE_PAIRS = {"LFW": 6000, "CFP_FP": 7000, "AgeDB30": 6000}   # dataloader.py:22-160


class SyntheticPairs:
    """Stand-in for the reference's LFW / CFP_FP / AgeDB30 datasets (``dataloader.py:22-160``): ``folds``,
    ``flags`` and items of four crops.  A pair's crops are a function of (seed, pair index) only, so any batch
    can be produced on its own, at any batch size, on the device that will embed it."""

    def __init__(self, pairs: int, seed: int = 0, noise: float = 0.45, device=None):
        ps = synthetic.PairSet(pairs)
        self.folds, self.flags = ps.folds, ps.flags
        self.pairs, self.seed, self.noise = pairs, int(seed), float(noise)
        self.device = torch.device(device) if device is not None else torch.device("cpu")

    def __len__(self) -> int:
        return self.pairs

    CHUNK = 10     # pairs are generated in fixed chunks, so a pair's crops do not depend on the batch size it is served in

    def _chunk(self, c: int):
        first = c * self.CHUNK
        n = min(self.CHUNK, self.pairs - first)
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 1_000_003 + c)
        up = lambda t: t.repeat_interleave(8, 2).repeat_interleave(8, 3)    # noqa: E731
        t1 = up(torch.rand((n, 3, 14, 14), generator=g, device=self.device) * 2 - 1)
        t2 = up(torch.rand((n, 3, 14, 14), generator=g, device=self.device) * 2 - 1)
        same = torch.tensor(self.flags[first:first + n], device=self.device).view(-1, 1, 1, 1) == 1
        left = (t1 + self.noise * torch.randn(t1.shape, generator=g, device=self.device)).clamp_(-1, 1)
        right = (torch.where(same, t1, t2) + self.noise * torch.randn(t1.shape, generator=g, device=self.device)).clamp_(-1, 1)
        return left, right

    def batch(self, start: int, size: int):
        """[L, flip(L), R, flip(R)] for pairs start .. start+size-1, f32 [size,3,112,112]."""
        end = min(start + size, self.pairs)
        parts = [self._chunk(c) for c in range(start // self.CHUNK, (end - 1) // self.CHUNK + 1)]
        lo = start - (start // self.CHUNK) * self.CHUNK
        left = torch.cat([p[0] for p in parts])[lo:lo + end - start].contiguous()
        right = torch.cat([p[1] for p in parts])[lo:lo + end - start].contiguous()
        return [left, torch.flip(left, dims=[3]), right, torch.flip(right, dims=[3])]   # cv2.flip(img, 1): dataloader.py:60


class PairLoader:
    """``DataLoader(dataset, batch_size, shuffle=False)`` over a SyntheticPairs set (``test_lfw_agedb_cfpfp.py:43``)."""

    def __init__(self, dataset: SyntheticPairs, batch_size: int):
        self.dataset, self.batch_size = dataset, int(batch_size)

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for start in range(0, len(self.dataset), self.batch_size):
            yield self.dataset.batch(start, self.batch_size)


def load_test_data(batch_size, dataset="Faces_emore", sizes=None, seed=0, noise=0.45, device=None, real=None, workers=2):
    """(dataloaders, dataset_sizes, dataset) keyed by phase, as ``test_lfw_agedb_cfpfp.py:22-55`` returns them.
    ``real``: {phase: (root, pair list)} for phases read from disk (``:27-47``: the reference's dataset classes, its
    transform, ``DataLoader(..., shuffle=False, num_workers=2)``); the other phases are synthetic."""
    from .data_set import dataloader as dl
    sizes = dict(PHASE_PAIRS if sizes is None else sizes)
    real = dict(real or {})
    classes = {"LFW": dl.LFW, "CFP_FP": dl.CFP_FP, "AgeDB30": dl.AgeDB30}
    sets, dataloaders = {}, {}
    for i, (name, n) in enumerate(sizes.items()):
        if name in real:
            root, file_list = real[name]
            sets[name] = classes[name](root, file_list, transform=dl.test_transform)
            dataloaders[name] = torch.utils.data.DataLoader(sets[name], batch_size=batch_size, shuffle=False, num_workers=workers)
        else:
            sets[name] = SyntheticPairs(n, seed=seed + 17 * i, noise=noise, device=device)
            dataloaders[name] = PairLoader(sets[name], batch_size)
    dataset_sizes = {name: len(ds) for name, ds in sets.items()}
    print("training and validation data loaded")
    return dataloaders, dataset_sizes, sets


def load_checkpoint(path: str):
    """The reference's ``torch.load(path)['net_state_dict']`` (``:62``), restricted to tensors: a plain state_dict,
    the training wrapper ``{'iters', 'net_state_dict'}`` and ``module.``-prefixed keys are all accepted by
    ``MobileFaceNet.load_state_dict``."""
    return torch.load(path, map_location="cpu", weights_only=True)


def run(model, device, batch_size=50, sizes=None, seed=0, noise=0.45, method="cos_distance", out=sys.stdout, real=None, workers=2):
    """The loop of ``test_lfw_agedb_cfpfp.py:74-78``.  Returns {phase: (ACCs[10], threshold[10])}."""
    dataloaders, _, dataset = load_test_data(batch_size, sizes=sizes, seed=seed, noise=noise, device=device, real=real, workers=workers)
    model.eval()
    results = {}
    for phase in dataloaders:
        featureLs, featureRs = getFeature(model, dataloaders[phase], device, flip=True)
        ACCs, threshold = evaluation_10_fold(featureLs, featureRs, dataset[phase], method=method)
        print("Epoch {}/{}，{} average acc:{:.4f} average threshold:{:.4f}".format(1, 1, phase, np.mean(ACCs) * 100,
                                                                                   np.mean(threshold)), file=out)
        results[phase] = (ACCs, threshold)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ckpt", help="checkpoint: state_dict or {'net_state_dict': ...} (Train_mobile.py:167-176)")
    ap.add_argument("--seed", type=int, default=1, help="without --ckpt: seed of the random weights")
    ap.add_argument("--batch-size", type=int, default=50)
    ap.add_argument("--data-seed", type=int, default=0)
    ap.add_argument("--noise", type=float, default=0.45)
    ap.add_argument("--pairs", type=int, nargs=3, metavar=("LFW", "CFP_FP", "AGEDB30"), default=None)
    ap.add_argument("--method", default="cos_distance", choices=["cos_distance", "l2_distance"])
    ap.add_argument("--lfw", nargs=2, metavar=("ROOT", "PAIRS"), help="aligned 112x112 LFW images and pairs.txt (:27-29)")
    ap.add_argument("--cfp-fp", nargs=2, metavar=("ROOT", "PAIRS"), help="CFP-FP images and cfp_fp_pair.txt (:35-37)")
    ap.add_argument("--agedb30", nargs=2, metavar=("ROOT", "PAIRS"), help="AgeDB-30 images and agedb_30_pair.txt (:39-41)")
    ap.add_argument("--workers", type=int, default=2, help="DataLoader workers for the phases read from disk")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("this harness needs a HIP device ('cuda:0'); there is no CPU path")
    device = torch.device("cuda:0")
    model = MobileFaceNet(512)
    if args.ckpt:
        model.load_state_dict(load_checkpoint(args.ckpt))
    else:
        sd = synthetic.make_state_dict(args.seed, 512)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(device)
    sizes = dict(zip(PHASE_PAIRS, args.pairs)) if args.pairs else None
    real = {k: tuple(v) for k, v in (("LFW", args.lfw), ("CFP_FP", args.cfp_fp), ("AgeDB30", args.agedb30)) if v}
    run(model, device, args.batch_size, sizes, args.data_seed, args.noise, args.method, real=real, workers=args.workers)


if __name__ == "__main__":
    main()
