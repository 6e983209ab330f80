"""Gallery store on disk: save -> load reproduces every search result bit for bit; per-rank loading,
incremental enrolment, and conversion from / to the reference's facebank.pth + names.npy pair."""
import json
import os

import numpy as np
import pytest
import torch

from face_recognition_amd import gallery_store as gs
from face_recognition_amd import synthetic
from face_recognition_amd.facebank import Gallery, merge_topk

pytestmark = pytest.mark.gpu


def _names(n):
    return [""] + [f"id{i}" for i in range(1, n + 1)]


def test_save_load_round_trip_is_bit_identical(device, tmp_path):
    N, Q = 10007, 300                                  # ragged: not a multiple of the shard or tile size
    rows = synthetic.unit_rows(N, 512, seed=3)
    q, truth = synthetic.planted_queries(rows, Q, seed=5, target_cos=0.7)
    g = Gallery(rows, store="f16", device=device)
    d0, i0 = g.search(torch.from_numpy(q), k=5)
    path = str(tmp_path / "store")
    meta = gs.save(path, g, _names(N), shard_rows=3000)
    assert meta["rows"] == N and [s["rows"] for s in meta["shards"]] == [3000, 3000, 3000, 1007]
    assert os.path.getsize(os.path.join(path, "rows_00003.bin")) == 1007 * 512 * 2
    g2, ids, names = gs.load(path, device=device, chunk_rows=1111)
    d1, i1 = g2.search(torch.from_numpy(q), k=5)
    assert torch.equal(d0, d1) and torch.equal(i0, i1)
    assert ids.tolist() == list(range(1, N + 1)) and names[0] == "" and names[7] == "id7"
    planted = truth >= 0
    assert (i1[:, 0].cpu().numpy()[planted] == truth[planted]).all()
    with pytest.raises(RuntimeError, match="already holds"):
        gs.save(path, g, _names(N))


def test_ranks_load_their_rows_and_merge_to_the_whole(device, tmp_path):
    N, Q, k = 5000, 64, 5
    rows = synthetic.unit_rows(N, 512, seed=8)
    q = torch.from_numpy(synthetic.planted_queries(rows, Q, seed=9, target_cos=0.6)[0])
    path = str(tmp_path / "store")
    gs.save(path, rows, _names(N), shard_rows=1700, device=device)
    whole, _, _ = gs.load(path, device=device)
    dw, iw = whole.search(q, k)
    parts = [gs.load(path, device=device, rank=r, world=3) for r in range(3)]
    assert sum(p[0].n for p in parts) == N and [p[0].index_base for p in parts][0] == 0
    dd, ii = zip(*[p[0].search(q, k) for p in parts])
    dm, im = merge_topk(torch.stack(dd), torch.stack(ii), k)
    assert torch.equal(dm, dw) and torch.equal(im, iw)
    assert torch.equal(torch.cat([p[1] for p in parts]), torch.arange(1, N + 1, dtype=torch.int32))


def test_append_enrols_new_identities(device, tmp_path):
    N, M = 2000, 300
    rows = synthetic.unit_rows(N + M, 512, seed=12)
    path = str(tmp_path / "store")
    gs.save(path, rows[:N], _names(N), shard_rows=1500, device=device)
    before = {f: os.path.getmtime(os.path.join(path, f)) for f in os.listdir(path) if f.startswith("rows_")}
    meta = gs.append(path, rows[N:], new_names=[f"new{i}" for i in range(M)], shard_rows=1500, device=device)
    assert meta["rows"] == N + M and len(meta["shards"]) == 3
    assert all(os.path.getmtime(os.path.join(path, f)) == t for f, t in before.items())    # old shards untouched
    g, ids, names = gs.load(path, device=device)
    ref = Gallery(rows, store="f16", device=device)
    q = torch.from_numpy(rows[N:N + 50] + 0.01 * synthetic.unit_rows(50, 512, seed=1))
    d1, i1 = g.search(q, 3)
    d2, i2 = ref.search(q, 3)
    assert torch.equal(d1, d2) and torch.equal(i1, i2)
    assert i1[:, 0].cpu().tolist() == list(range(N, N + 50))
    assert [names[ids[r]] for r in i1[:3, 0].cpu().tolist()] == ["new0", "new1", "new2"]
    # several rows of one identity
    gs.append(path, rows[:2], ids=[5, 5], device=device)
    _, ids, _ = gs.load(path, device=device)
    assert ids[-2:].tolist() == [5, 5]
    with pytest.raises(RuntimeError, match="new names"):
        gs.append(path, rows[:2], device=device)


def test_reference_file_pair_round_trip(device, tmp_path, golden_dir):
    """The shipped facebank's contents (5 x 512 fp32, names ['', 'Howard', ...]) through the reference's formats."""
    z = np.load(os.path.join(golden_dir, "facebank_match.npz"))
    fb, names = torch.from_numpy(z["facebank"]), [str(x) for x in z["names"]]
    pth, npy = str(tmp_path / "facebank.pth"), str(tmp_path / "names.npy")
    torch.save(fb, pth)
    np.save(npy, np.array(names))
    path = str(tmp_path / "store")
    meta = gs.import_reference(pth, npy, path, device=device)
    assert meta["rows"] == 5 and meta["names"] == 6
    g, ids, got_names = gs.load(path, device=device)
    assert got_names == names and ids.tolist() == [1, 2, 3, 4, 5]
    d, i = g.search(fb, k=1)                                    # every identity finds itself
    assert i[:, 0].cpu().tolist() == [0, 1, 2, 3, 4] and float(d.max()) < 1e-3
    gs.export_reference(path, str(tmp_path / "out.pth"), str(tmp_path / "out.npy"))
    back = torch.load(str(tmp_path / "out.pth"), weights_only=True)
    assert back.shape == fb.shape and torch.allclose(back, fb, atol=1e-3)        # 16-bit storage
    assert torch.equal(back, fb.half().float())
    assert list(np.load(str(tmp_path / "out.npy"))) == names


def test_store_errors(device, tmp_path):
    rows = synthetic.unit_rows(10, 512, seed=1)
    with pytest.raises(RuntimeError, match="unknown slot"):
        gs.save(str(tmp_path / "a"), rows, ["x"] * 11, device=device)
    with pytest.raises(RuntimeError, match="no name"):
        gs.save(str(tmp_path / "b"), rows, ["", "only"], device=device)
    with pytest.raises(RuntimeError, match="16-bit"):
        gs.save(str(tmp_path / "c"), Gallery(rows, store="f32", device=device), _names(10))
    path = str(tmp_path / "d")
    gs.save(path, rows, _names(10), device=device)
    meta = json.load(open(os.path.join(path, "meta.json")))
    meta["version"] = 99
    json.dump(meta, open(os.path.join(path, "meta.json"), "w"))
    with pytest.raises(RuntimeError, match="version"):
        gs.load(path, device=device)
