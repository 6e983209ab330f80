"""Drop-in for the reference's ``facebank.py`` plus the match step its demos repeat inline.

``load_facebank`` / ``prepare_facebank`` keep the reference's signatures, return types and
on-disk format (``facebank.pth`` + ``names.npy``, ``facebank.py:33-83``).  ``match_facebank`` is
the ten lines of ``cam_demo.py:96-105`` (= ``Video_demo.py:99-108``, ``MTCNN_MobileFaceNet.py:76-85``)
as one call; ``topk_search`` / ``Gallery`` extend it to top-k over galleries of up to ~10^8 rows.
All distance work runs in the HIP library; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from pathlib import Path
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .face_model import MobileFaceNet, l2_norm, tta_combine  # noqa: F401  (re-exported like the reference)


def listdir_nohidden(path):
    for f in os.listdir(path):
        if not f.startswith("."):
            yield f


def _imread_bgr(image_path: str) -> np.ndarray:
    """uint8 HWC BGR, the layout cv2.imread returns (facebank.py:47)."""
    try:
        import cv2  # the reference's reader, used when present
        return cv2.imread(image_path)
    except ImportError:
        from PIL import Image
        with Image.open(image_path) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])


def test_transform(img_bgr_u8: np.ndarray, device) -> torch.Tensor:
    """ToTensor + Normalize([.5]*3, [.5]*3) (facebank.py:22-24): uint8 HWC -> f32 CHW in [-1,1]."""
    t = torch.from_numpy(np.ascontiguousarray(img_bgr_u8)).to(device)
    return ((t.permute(2, 0, 1).to(torch.float32) / 255.0) - 0.5) / 0.5


class Gallery:
    """A facebank resident on one HIP device.

    ``store='f32'`` keeps rows in fp32 and evaluates sum((q-g)^2) exactly (the demos' arithmetic;
    any feature width).  ``store='f16'`` / ``'bf16'`` packs 512-wide rows for the MFMA search.
    ``index_base`` is added to reported indices (row offset of a shard inside a larger gallery).
    """

    def __init__(self, rows: Union[torch.Tensor, np.ndarray, None] = None, n: Optional[int] = None, d: int = 512,
                 store: str = "f16", device: Union[str, torch.device, None] = None, index_base: int = 0):
        lib = _lib.load()
        if rows is not None:
            if isinstance(rows, np.ndarray):
                rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32))
            if rows.dim() != 2:
                raise RuntimeError(f"gallery must be [N,D], got {tuple(rows.shape)}")
            rows = rows.detach().to(torch.float32).contiguous()
            n, d = int(rows.shape[0]), int(rows.shape[1])
            if device is None and rows.is_cuda:
                device = rows.device
        if n is None or n <= 0:
            raise RuntimeError("empty gallery")
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("Gallery lives on the HIP device ('cuda'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device, self.n, self.d, self.store, self.index_base = device, n, d, store, int(index_base)
        h = C.c_void_p()
        with torch.cuda.device(device):
            if rows is not None and rows.is_cuda and rows.device != device:
                rows = rows.to(device)
            if rows is not None and rows.is_cuda:
                # fb_create packs on a stream of its own: the rows must have been written by then
                torch.cuda.current_stream(device).synchronize()
            _lib.check(lib.fb_create(C.byref(h), device.index, rows.data_ptr() if rows is not None else None,
                                     1 if (rows is not None and rows.is_cuda) else 0, n, d,
                                     _lib.DTYPE_BY_NAME[store], self.index_base), "fb_create")
        self._h = h.value

    def fill(self, row0: int, rows: torch.Tensor) -> None:
        """Write rows [row0, row0+len(rows)).  Device rows are packed on the current stream without waiting (later
        searches on that stream see them); host rows are copied synchronously, after the searches already queued on the
        current stream have finished (they read the rows being replaced)."""
        rows = rows.detach().to(torch.float32).contiguous()
        with torch.cuda.device(self.device):
            if rows.is_cuda:
                if rows.device != self.device:
                    rows = rows.to(self.device)
                rows.record_stream(torch.cuda.current_stream(self.device))     # keep the block until the pack kernel has read it
                _lib.check(_lib.load().fb_fill_async(self._h, int(row0), rows.data_ptr(), int(rows.shape[0]),
                                                     _lib.current_stream_ptr(self.device)), "fb_fill_async")
            else:
                # fb_fill copies and packs on a stream of the handle's own: a search queued earlier on the caller's stream
                # and still running would read rows while they are overwritten, so let that stream drain first
                torch.cuda.current_stream(self.device).synchronize()
                _lib.check(_lib.load().fb_fill(self._h, int(row0), rows.data_ptr(), 0, int(rows.shape[0])), "fb_fill")

    def reserve(self, max_queries: int) -> None:
        """Pre-size internal scratch for searches of up to ``max_queries`` queries (needed before graph capture)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().fb_reserve(self._h, int(max_queries)), "fb_reserve")

    def search(self, queries: torch.Tensor, k: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
        """(dist f32 [Q,k] ascending, idx int32 [Q,k]) on the device; ties -> lowest index; empty slots (inf, -1)."""
        if queries.dim() != 2 or queries.shape[1] != self.d:
            raise RuntimeError(f"queries must be [Q,{self.d}], got {tuple(queries.shape)}")
        if queries.shape[0] == 0:
            raise RuntimeError("no queries")
        q = queries.detach().to(self.device, torch.float32).contiguous()
        dist = torch.empty((q.shape[0], k), dtype=torch.float32, device=self.device)
        idx = torch.empty((q.shape[0], k), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().fb_search_l2_topk(self._h, q.data_ptr(), q.shape[0], k, dist.data_ptr(),
                                                     idx.data_ptr(), _lib.current_stream_ptr(self.device)),
                       "fb_search_l2_topk")
        return dist, idx

    def close(self) -> None:
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):   # the C side frees on the handle's device
                _lib.load().fb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk(dist_parts: torch.Tensor, idx_parts: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [R,Q,k] per-shard candidate lists into the k best per query (device tensors)."""
    if dist_parts.dim() != 3 or dist_parts.shape != idx_parts.shape or dist_parts.shape[2] != k:
        raise RuntimeError(f"merge_topk: expected matching [R,Q,{k}] tensors")
    d = dist_parts.contiguous().float()
    i = idx_parts.contiguous().to(torch.int32)
    r, q, _ = d.shape
    dist = torch.empty((q, k), dtype=torch.float32, device=d.device)
    idx = torch.empty((q, k), dtype=torch.int32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().fb_merge_topk(d.data_ptr(), i.data_ptr(), r, q, k, dist.data_ptr(), idx.data_ptr(),
                                             _lib.current_stream_ptr(d.device)), "fb_merge_topk")
    return dist, idx


# (id(tensor), store) -> (weak reference to the tensor, its state when packed, Gallery).  An entry is
# only valid for the very tensor OBJECT it was built from: a freed facebank's address (and _version 0)
# is routinely handed to the next same-shaped tensor by the caching allocator, so neither identifies it.
_gallery_cache = {}


def _tensor_state(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device), t.dtype)


def _as_gallery(targets, store: str) -> Gallery:
    """Device copy of a facebank tensor, cached per tensor object.

    The copy is rebuilt when the tensor object differs or its ``_version`` moved (in-place torch ops).
    Writes that bypass the version counter (``.data``, kernels outside torch) are not seen: build a
    ``Gallery`` yourself and ``fill`` it when rows change that way.
    """
    if isinstance(targets, Gallery):
        return targets
    for k in [k for k, (ref, _, _) in _gallery_cache.items() if ref() is None]:   # owners that died
        _gallery_cache.pop(k)[2].close()
    key = (id(targets), store)
    hit = _gallery_cache.get(key)
    if hit is not None:
        ref, state, g = hit
        if ref() is targets and state == _tensor_state(targets):
            return g
        _gallery_cache.pop(key)[2].close()
    if len(_gallery_cache) >= 4:
        _gallery_cache.pop(next(iter(_gallery_cache)))[2].close()
    dev = targets.device if targets.is_cuda else None
    g = Gallery(targets, store=store, device=dev)
    _gallery_cache[key] = (weakref.ref(targets), _tensor_state(targets), g)
    return g


def topk_search(queries: torch.Tensor, gallery: Union[torch.Tensor, Gallery], k: int = 5, store: Optional[str] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest gallery rows per query by squared L2: (dist f32 [Q,k], idx int64 [Q,k]), best first."""
    if store is None:
        n = gallery.n if isinstance(gallery, Gallery) else gallery.shape[0]
        d = gallery.d if isinstance(gallery, Gallery) else gallery.shape[1]
        store = "f16" if (n > 4096 and d == 512) else "f32"
    g = _as_gallery(gallery, store)
    dist, idx = g.search(queries, k)
    return dist, idx.to(torch.int64)


def match_facebank(source_embs: torch.Tensor, targets: Union[torch.Tensor, Gallery], threshold: float = 60
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The demos' match step (cam_demo.py:97-105).

    Returns ``(minimum f32[Q], min_idx int64[Q], score_100 f32[Q])``: squared L2 distance to the
    nearest facebank row, its index (-1 when ``minimum > (threshold-156)/(-80)``; ``names[idx+1]`` is
    then the reference's '' = unknown slot), and ``clamp(-80*minimum+156, 0, 100)``.
    """
    g = _as_gallery(targets, "f32")
    if g.store == "f32" and g.n > 0:          # the demos' case: distances, nearest row, threshold and score in two launches
        if source_embs.dim() != 2 or source_embs.shape[1] != g.d or source_embs.shape[0] == 0:
            raise RuntimeError(f"queries must be [Q,{g.d}] with Q > 0, got {tuple(source_embs.shape)}")
        q = source_embs.detach().to(g.device, torch.float32).contiguous()
        minimum = torch.empty(q.shape[0], dtype=torch.float32, device=g.device)
        min_idx = torch.empty(q.shape[0], dtype=torch.int64, device=g.device)
        score = torch.empty_like(minimum)
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().fb_match_nearest(g._h, q.data_ptr(), q.shape[0], float(threshold), minimum.data_ptr(),
                                                    min_idx.data_ptr(), score.data_ptr(), _lib.current_stream_ptr(g.device)),
                       "fb_match_nearest")
        return minimum, min_idx, score
    dist, idx = g.search(source_embs, 1)
    minimum = dist[:, 0].contiguous()
    min_idx = idx[:, 0].to(torch.int64).contiguous()
    score = torch.empty_like(minimum)
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().fb_threshold_score(minimum.data_ptr(), min_idx.data_ptr(), score.data_ptr(),
                                                  minimum.shape[0], float(threshold),
                                                  _lib.current_stream_ptr(g.device)), "fb_threshold_score")
    return minimum, min_idx, score


def prepare_facebank(model: MobileFaceNet, path: str = "facebank", tta: bool = True,
                     align_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None):
    """Enrol every sub-directory of ``path`` as one identity (facebank.py:33-77).

    Per identity: mean over its images of the (flip-TTA) embedding, NOT re-normalised (:68);
    ``names = ['', dir names...]`` (:36,70); writes ``facebank.pth`` and ``names.npy`` (:74-75).
    Images must already be 112x112 aligned crops unless ``align_fn`` (uint8 HWC BGR image ->
    112x112x3 crop) is given: detection and alignment stay with the reference's MTCNN code (:49-55).
    """
    model.eval()
    device = next(model.parameters()).device
    embeddings = []
    names = [""]
    data_path = Path(path)
    for doc in data_path.iterdir():
        if doc.is_file():
            continue
        crops = []      # uint8 HWC BGR crops; ToTensor + Normalize happen inside the first kernel
        for files in listdir_nohidden(doc):
            img = _imread_bgr(os.path.join(doc, files))
            if img is None:
                raise RuntimeError(f"cannot read image {os.path.join(doc, files)}")
            if img.shape != (112, 112, 3):
                if align_fn is None:
                    raise RuntimeError(f"{os.path.join(doc, files)} is {img.shape}, not a 112x112 aligned crop; "
                                       "pass align_fn (e.g. the reference's MTCNN + Face_alignment)")
                img = align_fn(img)
            crops.append(np.ascontiguousarray(img, dtype=np.uint8))
        if len(crops) == 0:
            continue
        x = torch.from_numpy(np.stack(crops)).to(device)
        with torch.no_grad():
            embs = model.forward_u8(x, tta=tta)
        embeddings.append(embs.mean(0, keepdim=True))
        names.append(doc.name)
    embeddings = torch.cat(embeddings)
    names = np.array(names)
    torch.save(embeddings, os.path.join(path, "facebank.pth"))
    np.save(os.path.join(path, "names"), names)
    return embeddings, names


def load_facebank(path: str = "facebank", map_location=None):
    """(Tensor[N,512], ndarray[N+1] of names) from ``path/facebank.pth`` + ``path/names.npy`` (facebank.py:79-83)."""
    data_path = Path(path)
    embeddings = torch.load(data_path / "facebank.pth", map_location=map_location)
    names = np.load(data_path / "names.npy")
    return embeddings, names
