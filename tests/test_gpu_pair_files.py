"""Verification pairs read from files, both routes to the GPU: the reference's float items through getFeature, and raw
uint8 crops through getFeature_u8 (transform and mirrored copy inside the first kernel, copies under forwards)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from face_recognition_amd import Evaluation
from face_recognition_amd.data_set import dataloader as dl
from oracle import mobilefacenet_ref
from tests.conftest import cosine_rows

pytestmark = pytest.mark.gpu


def test_pair_files_float_and_raw_routes(tmp_path, hip_model, device, state_dict_np):
    rng = np.random.default_rng(12)
    pairs, lines = 22, []
    os.makedirs(tmp_path / "img")
    for i in range(pairs):
        for side in "lr":
            img = np.repeat(np.repeat(rng.integers(0, 256, (14, 14, 3), dtype=np.uint8), 8, 0), 8, 1)
            Image.fromarray(img, "RGB").save(tmp_path / "img" / f"{i}_{side}.png")
        lines.append(f"img/{i}_l.png img/{i}_r.png {1 if i % 2 else -1}")
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    ds = dl.AgeDB30(str(tmp_path), str(tmp_path / "list.txt"), transform=dl.test_transform)
    loader = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, num_workers=0)
    fl, fr = Evaluation.getFeature(hip_model, loader, device, flip=True)
    assert tuple(fl.shape) == (pairs, 512) and fl.device.type == "cpu"
    # the float route against the oracle on the same items
    item = ds[3]
    embed = lambda x: mobilefacenet_ref.forward(state_dict_np, x)      # noqa: E731
    a, b = embed(item[0][None]), embed(item[1][None])
    want = torch.nn.functional.normalize(a + b)
    assert cosine_rows(fl[3:4].numpy(), want.numpy()).min() >= 0.9999
    # the raw route: same features to the bar, left / right split and order kept, for several batch sizes
    for bs in (5, 22, 1):
        ul, ur = Evaluation.getFeature_u8(hip_model, dl.u8_pair_batches(ds, bs), device, flip=True)
        assert tuple(ul.shape) == (pairs, 512)
        assert cosine_rows(ul.numpy(), fl.numpy()).min() >= 0.9999 and cosine_rows(ur.numpy(), fr.numpy()).min() >= 0.9999
    nl, _ = Evaluation.getFeature_u8(hip_model, dl.u8_pair_batches(ds, 7), device, flip=False, keep_on_device=True)
    pl, _ = Evaluation.getFeature(hip_model, loader, device, flip=False)
    assert nl.is_cuda and cosine_rows(nl.cpu().numpy(), pl.numpy()).min() >= 0.9999
    assert Evaluation.getFeature_u8(hip_model, iter(()), device) == (None, None)
    with pytest.raises(ValueError):
        Evaluation.getFeature_u8(hip_model, dl.u8_pair_batches(ds, 22), device, max_batch=32)
