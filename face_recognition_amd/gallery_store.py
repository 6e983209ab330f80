"""Facebank on disk, sized for 100 M identities.

The reference keeps its facebank as one fp32 tensor plus a names array (``facebank.pth`` /
``names.npy``, ``facebank.py:72-75`` written, ``:79-83`` read).  That is 205 GB at 100 M rows and has to
be converted on every load.  A store directory holds instead what the search kernel reads, in shards:

    meta.json            {"version", "d", "dtype", "rows", "shards": [{"rows", "row0"}...], "names"}
    rows_00000.bin       16-bit rows [n, d], little endian, exactly the packed gallery rows
    negh_00000.f32       -|g|^2/2 per row (of the rounded values)
    ids_00000.i32        identity id of every row (index into names.json); several rows may share one
    names.json           identity names; entry 0 is '' ("unknown"), as in the reference's names.npy

Loading copies shards straight into HBM (``fb_import_packed``): no conversion, and search results
are bit-identical to those of the gallery that was saved.  A rank of a multi-GPU job loads only the rows
``sharded.shard_bounds`` assigns to it.  ``append`` adds identities as new shards without touching
the old ones.  ``import_reference`` / ``export_reference`` convert from and to the reference's
file pair.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .facebank import Gallery

VERSION = 1
_NP16 = {"f16": np.float16, "bf16": np.uint16}      # bf16 rows are kept as raw 16-bit words


def _shard_file(path: str, kind: str, i: int) -> str:
    ext = {"rows": "bin", "negh": "f32", "ids": "i32"}[kind]
    return os.path.join(path, f"{kind}_{i:05d}.{ext}")


def read_meta(path: str) -> dict:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("version") != VERSION:
        raise RuntimeError(f"{path}: store version {meta.get('version')!r}, this build reads {VERSION}")
    return meta


def _write_meta(path: str, meta: dict) -> None:
    tmp = os.path.join(path, "meta.json.tmp")
    with open(tmp, "w") as f:
        json.dump(meta, f, indent=1)
    os.replace(tmp, os.path.join(path, "meta.json"))      # shards first, meta last: a reader never sees half a shard


def read_names(path: str) -> List[str]:
    with open(os.path.join(path, "names.json")) as f:
        return json.load(f)


def _write_shards(path: str, first_shard: int, row0: int, gallery: Gallery, ids: np.ndarray, shard_rows: int) -> List[dict]:
    """Export the packed rows of ``gallery`` (all of them) as shards starting at global row ``row0``."""
    lib = _lib.load()
    entries = []
    n = gallery.n
    for k, lo in enumerate(range(0, n, shard_rows)):
        m = min(shard_rows, n - lo)
        rows = np.empty((m, gallery.d), dtype=_NP16[gallery.store])
        negh = np.empty(m, dtype=np.float32)
        with torch.cuda.device(gallery.device):
            torch.cuda.synchronize(gallery.device)
            _lib.check(lib.fb_export_packed(gallery._h, lo, m, rows.ctypes.data, 0, negh.ctypes.data, 0), "fb_export_packed")
        i = first_shard + k
        rows.tofile(_shard_file(path, "rows", i))
        negh.tofile(_shard_file(path, "negh", i))
        np.ascontiguousarray(ids[lo:lo + m], dtype=np.int32).tofile(_shard_file(path, "ids", i))
        entries.append({"rows": int(m), "row0": int(row0 + lo)})
    return entries


def _as_ids(ids, n: int) -> np.ndarray:
    ids = np.arange(1, n + 1, dtype=np.int32) if ids is None else np.asarray(ids, dtype=np.int64)
    if ids.shape != (n,):
        raise RuntimeError(f"ids must have one entry per row ({n}), got shape {ids.shape}")
    if ids.size and (ids.min() < 0 or ids.max() >= 2 ** 31):
        raise RuntimeError("ids must fit int32 and be >= 0")
    return ids.astype(np.int32)


def save(path: str, embeddings: Union[torch.Tensor, np.ndarray, Gallery], names: Sequence[str], ids=None,
         store: str = "f16", shard_rows: int = 1 << 22, device=None) -> dict:
    """Create a store at ``path`` from fp32 embeddings [N,512] (or a packed ``Gallery``).

    ``names[0]`` must be '' (the reference's "unknown" slot); ``ids[r]`` is the index into ``names`` of
    row r (default: row r is identity r + 1, the reference's one-row-per-identity layout)."""
    if os.path.exists(os.path.join(path, "meta.json")):
        raise RuntimeError(f"{path} already holds a store; use append()")
    names = [str(x) for x in names]
    if not names or names[0] != "":
        raise RuntimeError("names[0] must be '' (the unknown slot, as in the reference's names.npy)")
    g = embeddings if isinstance(embeddings, Gallery) else Gallery(embeddings, store=store, device=device)
    if g.store not in _NP16:
        raise RuntimeError("a store holds 16-bit galleries (store='f16' or 'bf16')")
    ids = _as_ids(ids, g.n)
    if ids.size and ids.max() >= len(names):
        raise RuntimeError(f"id {int(ids.max())} has no name ({len(names)} names)")
    os.makedirs(path, exist_ok=True)
    shards = _write_shards(path, 0, 0, g, ids, int(shard_rows))
    with open(os.path.join(path, "names.json"), "w") as f:
        json.dump(names, f)
    meta = {"version": VERSION, "d": g.d, "dtype": g.store, "rows": g.n, "shards": shards, "names": len(names)}
    _write_meta(path, meta)
    return meta


def append(path: str, embeddings: Union[torch.Tensor, np.ndarray], new_names: Sequence[str] = (), ids=None,
           shard_rows: int = 1 << 22, device=None) -> dict:
    """Enrol more rows: they become new shards; existing files are not rewritten (except the two small
    json files).  ``new_names`` are appended to names.json; ``ids`` default to one new identity per row,
    in which case ``new_names`` must name them."""
    meta = read_meta(path)
    names = read_names(path)
    g = Gallery(embeddings, store=meta["dtype"], device=device)
    if g.d != meta["d"]:
        raise RuntimeError(f"store holds {meta['d']}-wide rows, got {g.d}")
    new_names = [str(x) for x in new_names]
    if ids is None:
        if len(new_names) != g.n:
            raise RuntimeError(f"{g.n} new rows need {g.n} new names (or explicit ids)")
        ids = np.arange(len(names), len(names) + g.n)
    ids = _as_ids(ids, g.n)
    names = names + new_names
    if ids.size and ids.max() >= len(names):
        raise RuntimeError(f"id {int(ids.max())} has no name ({len(names)} names)")
    shards = _write_shards(path, len(meta["shards"]), meta["rows"], g, ids, int(shard_rows))
    with open(os.path.join(path, "names.json.tmp"), "w") as f:
        json.dump(names, f)
    os.replace(os.path.join(path, "names.json.tmp"), os.path.join(path, "names.json"))
    meta["shards"] += shards
    meta["rows"] += g.n
    meta["names"] = len(names)
    _write_meta(path, meta)
    return meta


def _overlaps(meta: dict, lo: int, hi: int):
    """(shard index, first row inside the shard, row count, offset inside [lo,hi)) of every shard part in [lo,hi)."""
    for i, sh in enumerate(meta["shards"]):
        a, b = max(lo, sh["row0"]), min(hi, sh["row0"] + sh["rows"])
        if a < b:
            yield i, a - sh["row0"], b - a, a - lo


def load(path: str, device=None, rank: int = 0, world: int = 1,
         chunk_rows: int = 1 << 20) -> Tuple[Gallery, torch.Tensor, List[str]]:
    """(gallery, ids int32 [n] on the host, names).  With ``world`` > 1 the gallery holds this rank's rows only
    and reports global row indices (``index_base``); ``ids`` covers the same rows."""
    from .sharded import shard_bounds
    meta = read_meta(path)
    lo, hi = shard_bounds(meta["rows"], world, rank)
    if hi <= lo:
        raise RuntimeError(f"rank {rank} of {world} gets no rows of a {meta['rows']}-row store")
    g = Gallery(None, n=hi - lo, d=meta["d"], store=meta["dtype"], device=device, index_base=lo)
    lib = _lib.load()
    ids = np.empty(hi - lo, dtype=np.int32)
    np16 = _NP16[meta["dtype"]]
    for i, first, cnt, off in _overlaps(meta, lo, hi):
        rows = np.memmap(_shard_file(path, "rows", i), dtype=np16, mode="r", shape=(meta["shards"][i]["rows"], meta["d"]))
        negh = np.memmap(_shard_file(path, "negh", i), dtype=np.float32, mode="r", shape=(meta["shards"][i]["rows"],))
        sid = np.memmap(_shard_file(path, "ids", i), dtype=np.int32, mode="r", shape=(meta["shards"][i]["rows"],))
        ids[off:off + cnt] = sid[first:first + cnt]
        for c in range(0, cnt, chunk_rows):                 # bounded host staging: a shard may be far larger than RAM
            m = min(chunk_rows, cnt - c)
            r = np.ascontiguousarray(rows[first + c:first + c + m])
            h = np.ascontiguousarray(negh[first + c:first + c + m])
            with torch.cuda.device(g.device):
                _lib.check(lib.fb_import_packed(g._h, off + c, m, r.ctypes.data, 0, h.ctypes.data, 0), "fb_import_packed")
    return g, torch.from_numpy(ids), read_names(path)


def import_reference(facebank_pth: str, names_npy: str, path: str, store: str = "f16", device=None) -> dict:
    """Build a store from the reference's file pair (facebank.py:72-75): embeddings [N,512] fp32, names[0] == ''."""
    emb = torch.load(facebank_pth, map_location="cpu", weights_only=True)
    names = np.load(names_npy, allow_pickle=False)
    if not isinstance(emb, torch.Tensor) or emb.dim() != 2:
        raise RuntimeError(f"{facebank_pth}: expected one [N,D] tensor")
    if len(names) != emb.shape[0] + 1:
        raise RuntimeError(f"{names_npy}: {len(names)} names for {emb.shape[0]} embeddings (want one more: the '' slot)")
    return save(path, emb.float(), [str(x) for x in names], store=store, device=device)


def export_reference(path: str, facebank_pth: str, names_npy: str) -> None:
    """Write the reference's file pair from a store whose rows are one per identity, in id order.
    The embeddings are the store's 16-bit values widened to fp32 (the store does not keep the originals)."""
    meta = read_meta(path)
    names = read_names(path)
    if meta["dtype"] != "f16":
        raise RuntimeError("export_reference reads f16 stores")
    out = np.empty((meta["rows"], meta["d"]), dtype=np.float32)
    ids = np.empty(meta["rows"], dtype=np.int32)
    for i, sh in enumerate(meta["shards"]):
        out[sh["row0"]:sh["row0"] + sh["rows"]] = np.fromfile(_shard_file(path, "rows", i), dtype=np.float16).reshape(sh["rows"], meta["d"])
        ids[sh["row0"]:sh["row0"] + sh["rows"]] = np.fromfile(_shard_file(path, "ids", i), dtype=np.int32)
    if not np.array_equal(ids, np.arange(1, meta["rows"] + 1)):
        raise RuntimeError("export_reference needs one row per identity in id order (the reference's layout)")
    torch.save(torch.from_numpy(out), facebank_pth)
    np.save(names_npy, np.array(names))
