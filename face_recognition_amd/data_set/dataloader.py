"""Pair lists on disk -> the items the verification harness embeds (reference: Mobilefacenet/data_set/dataloader.py:11-160).

``LFW``, ``CFP_FP`` and ``AgeDB30`` keep the reference's constructor ``(root, file_list, transform=None,
loader=img_loader)``, its attributes (``nameLs``, ``nameRs``, ``folds``, ``flags``) and its items
``[L, flip(L), R, flip(R)]``; they work with ``torch.utils.data.DataLoader`` as the reference's do
(``test_lfw_agedb_cfpfp.py:43``).  The three differ only in how a line of the pair list reads and in the pairs per
fold, so one class holds the rest.

Not the reference's behaviour, on purpose: a line that does not parse raises ``ValueError`` with its line number (the
reference keeps the previous line's names for it, or fails with an unbound name on the first line).

Images are decoded by cv2 when it is installed and by PIL otherwise (RGB -> BGR, grey -> three equal channels, as
``img_loader`` at :11-20 leaves them); whether both decoders give the same bytes for a jpg is not pinned here --
lossless files (png, bmp) are the same bytes by construction.

``u8_pair_batches`` is the route that skips the host-side float transform: raw uint8 crops for
``MobileFaceNet.forward_u8`` / ``streaming.EmbeddingStream``, which apply ToTensor + Normalize and the mirrored read
inside the first kernel.
"""
from __future__ import annotations

import os
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


def img_loader(path: str) -> Optional[np.ndarray]:
    """uint8 HWC BGR image, or None (with the reference's message) when the file cannot be read (:11-20)."""
    try:
        try:
            import cv2
        except ImportError:
            cv2 = None
        if cv2 is not None:
            img = cv2.imread(path)
            if img is None:
                raise IOError(path)
        else:
            from PIL import Image
            with Image.open(path) as im:
                img = np.asarray(im.convert("RGB") if im.mode not in ("L", "I;16", "1") else im.convert("L"))
            if img.ndim == 3:
                img = img[:, :, ::-1]                      # RGB -> BGR, what cv2.imread returns
        if img.ndim == 2:
            img = np.stack([img] * 3, 2)
        return np.ascontiguousarray(img)
    except (IOError, OSError):
        print("Cannot load image " + path)
        return None


def test_transform(img: np.ndarray) -> torch.Tensor:
    """``Compose([ToTensor(), Normalize((0.5,)*3, (0.5,)*3)])`` (test_lfw_agedb_cfpfp.py:24-26) on a uint8 HWC image,
    in torchvision's order of operations: f32 CHW / 255, then (t - 0.5) / 0.5."""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(0.5).div_(0.5)


def _flip_w(img: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(img[:, ::-1])              # cv2.flip(img, 1)


class _PairSet(torch.utils.data.Dataset):
    """Shared body of the three pair sets.  ``_parse(fields, i)`` -> (nameL, nameR, flag) for line i."""

    SEP: Optional[str] = " "
    SKIP = 0               # header lines of the pair list
    PER_FOLD = 600

    def __init__(self, root: str, file_list: str, transform: Optional[Callable] = None, loader: Callable = img_loader):
        self.root, self.file_list, self.transform, self.loader = root, file_list, transform, loader
        self.nameLs: List[str] = []
        self.nameRs: List[str] = []
        self.folds: List[int] = []
        self.flags: List[int] = []
        with open(file_list) as f:
            lines = f.read().splitlines()[self.SKIP:]
        for i, line in enumerate(lines):
            try:
                nameL, nameR, flag = self._parse(line.split(self.SEP), i)
            except (IndexError, ValueError) as e:
                raise ValueError(f"{file_list}: line {i + 1 + self.SKIP} does not read as a pair: {line!r}") from e
            self.nameLs.append(nameL)
            self.nameRs.append(nameR)
            self.folds.append(i // self.PER_FOLD)
            self.flags.append(flag)

    def _parse(self, p: Sequence[str], i: int) -> Tuple[str, str, int]:
        return p[0], p[1], int(p[2])

    def __len__(self) -> int:
        return len(self.nameLs)

    def pair_u8(self, index: int) -> Tuple[np.ndarray, np.ndarray]:
        """The two decoded images of a pair, uint8 HWC BGR."""
        out = []
        for name in (self.nameLs[index], self.nameRs[index]):
            img = self.loader(os.path.join(self.root, name))
            if img is None:
                raise IOError(f"pair {index}: cannot load {os.path.join(self.root, name)}")
            out.append(img)
        return out[0], out[1]

    def __getitem__(self, index: int):
        img_l, img_r = self.pair_u8(index)
        imglist = [img_l, _flip_w(img_l), img_r, _flip_w(img_r)]
        if self.transform is not None:
            return [self.transform(im) for im in imglist]
        return [torch.from_numpy(im) for im in imglist]


class LFW(_PairSet):
    """``pairs.txt``: a header line, then tab-separated ``name n1 n2`` (same person) or ``nameA n1 nameB n2``;
    files ``name/name_%04d.jpg``; 600 pairs per fold (:22-52)."""
    SEP, SKIP, PER_FOLD = "\t", 1, 600

    def _parse(self, p, i):
        if len(p) == 3:
            return f"{p[0]}/{p[0]}_{int(p[1]):04}.jpg", f"{p[0]}/{p[0]}_{int(p[2]):04}.jpg", 1
        if len(p) == 4:
            return f"{p[0]}/{p[0]}_{int(p[1]):04}.jpg", f"{p[2]}/{p[2]}_{int(p[3]):04}.jpg", -1
        raise ValueError(f"{len(p)} fields")


class CFP_FP(_PairSet):
    """``pathL pathR flag`` separated by single spaces, 700 pairs per fold (:72-96)."""
    SEP, SKIP, PER_FOLD = " ", 0, 700


class AgeDB30(_PairSet):
    """``pathL pathR flag`` separated by single spaces, 600 pairs per fold (:117-141)."""
    SEP, SKIP, PER_FOLD = " ", 0, 600


def u8_pair_batches(dataset: _PairSet, batch_size: int) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
    """(L, R) uint8 [b,H,W,3] host tensors for consecutive pairs: the input of ``forward_u8(..., tta=True)`` /
    ``EmbeddingStream(tta=True)``, whose result per crop is what ``getFeature(flip=True)`` computes from an item's
    ``[L, flip(L)]`` (Evaluation.py:50-52)."""
    for start in range(0, len(dataset), batch_size):
        pairs = [dataset.pair_u8(i) for i in range(start, min(start + batch_size, len(dataset)))]
        yield (torch.from_numpy(np.stack([p[0] for p in pairs])), torch.from_numpy(np.stack([p[1] for p in pairs])))
