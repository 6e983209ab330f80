"""Pair lists and images on disk -> the items the verification harness embeds
(face_recognition_amd/data_set/dataloader.py against oracle/dataloader_ref.py and hand-written expectations;
reference: Mobilefacenet/data_set/dataloader.py:11-160)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from face_recognition_amd.data_set import dataloader as dl
from oracle import dataloader_ref as ref

LFW_TEXT = "10\t300\nAaron_Eckhart\t1\t4\nAbel_Pacheco\t2\t3\nAaron_Eckhart\t1\tAbel_Pacheco\t2\nZico\t12\tAbel_Pacheco\t3\n"
CFP_TEXT = "cfp/a_1.png cfp/a_2.png 1\ncfp/a_1.png cfp/b_1.png -1\n"


def _write_images(root, names, seed=0, size=112, grey=()):
    rng = np.random.default_rng(seed)
    out = {}
    for n in names:
        os.makedirs(os.path.dirname(os.path.join(root, n)), exist_ok=True)
        if n in grey:
            img = rng.integers(0, 256, (size, size), dtype=np.uint8)
            Image.fromarray(img, "L").save(os.path.join(root, n))
            out[n] = np.stack([img] * 3, 2)
        else:
            img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)      # BGR as the loader returns it
            Image.fromarray(img[:, :, ::-1].copy(), "RGB").save(os.path.join(root, n))
            out[n] = img
    return out


def test_lfw_pair_list(tmp_path):
    f = tmp_path / "pairs.txt"
    f.write_text(LFW_TEXT)
    ds = dl.LFW(str(tmp_path), str(f))
    assert ds.nameLs == ["Aaron_Eckhart/Aaron_Eckhart_0001.jpg", "Abel_Pacheco/Abel_Pacheco_0002.jpg",
                         "Aaron_Eckhart/Aaron_Eckhart_0001.jpg", "Zico/Zico_0012.jpg"]
    assert ds.nameRs == ["Aaron_Eckhart/Aaron_Eckhart_0004.jpg", "Abel_Pacheco/Abel_Pacheco_0003.jpg",
                         "Abel_Pacheco/Abel_Pacheco_0002.jpg", "Abel_Pacheco/Abel_Pacheco_0003.jpg"]
    assert ds.flags == [1, 1, -1, -1] and ds.folds == [0, 0, 0, 0] and len(ds) == 4
    assert (ds.nameLs, ds.nameRs, ds.folds, ds.flags) == ref.parse_lfw(LFW_TEXT)
    # 6000 lines: folds 0..9 in blocks of 600, as the 10-fold protocol expects (Evaluation.py:77-79)
    big = "hdr\n" + "".join(f"P{i}\t1\t2\n" if i % 2 else f"P{i}\t1\tQ{i}\t1\n" for i in range(6000))
    f.write_text(big)
    ds = dl.LFW(str(tmp_path), str(f))
    assert (ds.nameLs, ds.nameRs, ds.folds, ds.flags) == ref.parse_lfw(big)
    assert ds.folds[599] == 0 and ds.folds[600] == 1 and ds.folds[-1] == 9 and sorted(set(ds.flags)) == [-1, 1]
    f.write_text("hdr\nA\t1\t2\nonly_two\t7\n")
    with pytest.raises(ValueError, match="line 3"):
        dl.LFW(str(tmp_path), str(f))


@pytest.mark.parametrize("cls,per_fold", [(dl.CFP_FP, 700), (dl.AgeDB30, 600)])
def test_space_separated_pair_lists(tmp_path, cls, per_fold):
    text = "".join(f"d{i % 7}/l_{i}.jpg d{i % 5}/r_{i}.jpg {1 if i % 3 else -1}\n" for i in range(10 * per_fold))
    f = tmp_path / "list.txt"
    f.write_text(text)
    ds = cls(str(tmp_path), str(f))
    assert (ds.nameLs, ds.nameRs, ds.folds, ds.flags) == ref.parse_space_list(text, per_fold)
    assert len(ds) == 10 * per_fold and ds.folds[per_fold - 1] == 0 and ds.folds[per_fold] == 1 and ds.folds[-1] == 9
    f.write_text("a.jpg b.jpg 1\na.jpg b.jpg x\n")
    with pytest.raises(ValueError, match="line 2"):
        cls(str(tmp_path), str(f))


def test_items_transform_and_loader(tmp_path):
    names = ["cfp/a_1.png", "cfp/a_2.png", "cfp/b_1.png"]
    imgs = _write_images(str(tmp_path), names, seed=3, grey=("cfp/b_1.png",))
    f = tmp_path / "cfp.txt"
    f.write_text(CFP_TEXT)
    raw = dl.CFP_FP(str(tmp_path), str(f))
    for i, (nl, nr) in enumerate([("cfp/a_1.png", "cfp/a_2.png"), ("cfp/a_1.png", "cfp/b_1.png")]):
        want = ref.item(imgs[nl], imgs[nr])
        got = raw[i]
        assert len(got) == 4 and all(g.dtype == torch.uint8 and np.array_equal(g.numpy(), w) for g, w in zip(got, want))
    ds = dl.CFP_FP(str(tmp_path), str(f), transform=dl.test_transform)
    got = ds[1]
    want = [ref.transform(w) for w in ref.item(imgs["cfp/a_1.png"], imgs["cfp/b_1.png"])]
    assert all(g.dtype == torch.float32 and tuple(g.shape) == (3, 112, 112) and np.array_equal(g.numpy(), w) for g, w in zip(got, want))
    assert float(got[0].min()) >= -1.0 and float(got[0].max()) <= 1.0
    # DataLoader(dataset, batch_size, shuffle=False) collates an item's four images into four batches (test_lfw...:43)
    det = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    assert len(det) == 4 and all(tuple(d.shape) == (2, 3, 112, 112) for d in det)
    assert torch.equal(det[1], torch.flip(det[0], dims=[3])) and torch.equal(det[3][1], got[3])
    # raw uint8 route
    (L, R), = list(dl.u8_pair_batches(raw, 8))
    assert L.dtype == torch.uint8 and tuple(L.shape) == (2, 112, 112, 3)
    assert np.array_equal(L[1].numpy(), imgs["cfp/a_1.png"]) and np.array_equal(R[1].numpy(), imgs["cfp/b_1.png"])
    assert [tuple(l.shape[:1]) for l, _ in dl.u8_pair_batches(raw, 1)] == [(1,), (1,)]
    # a missing file: the reference's message, and an error naming the pair instead of a None in the batch
    f.write_text("cfp/a_1.png cfp/nope.png 1\n")
    bad = dl.CFP_FP(str(tmp_path), str(f))
    assert dl.img_loader(str(tmp_path / "cfp/nope.png")) is None
    with pytest.raises(IOError, match="pair 0"):
        bad[0]


def test_harness_reads_pair_lists_from_disk(tmp_path, capsys):
    from face_recognition_amd import test_lfw_agedb_cfpfp as harness
    _write_images(str(tmp_path), ["cfp/a_1.png", "cfp/a_2.png", "cfp/b_1.png"], seed=4)
    f = tmp_path / "cfp.txt"
    f.write_text(CFP_TEXT)
    loaders, sizes, sets = harness.load_test_data(2, sizes={"LFW": 20, "CFP_FP": 0}, real={"CFP_FP": (str(tmp_path), str(f))}, workers=0)
    assert sizes == {"LFW": 20, "CFP_FP": 2} and isinstance(sets["CFP_FP"], dl.CFP_FP)
    det = next(iter(loaders["CFP_FP"]))
    assert len(det) == 4 and tuple(det[0].shape) == (2, 3, 112, 112) and det[0].dtype == torch.float32
    assert "training and validation data loaded" in capsys.readouterr().out
