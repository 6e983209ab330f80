"""hipGraph replay of forward (+TTA) + match equals the eager path."""
import numpy as np
import pytest
import torch

from face_recognition_amd import synthetic
from face_recognition_amd.facebank import match_facebank
from face_recognition_amd.graphed import GraphedRecognizer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tta", [False, True])
def test_graph_replay_equals_eager(hip_model, device, tta):
    bank = torch.from_numpy(synthetic.unit_rows(50, 512, seed=2)).to(device)
    rec = GraphedRecognizer(hip_model, bank, tta=tta, threshold=60)
    for B, seed in ((1, 1), (4, 2), (1, 3), (8, 4), (4, 5)):       # re-uses captured graphs for repeated sizes
        x = torch.from_numpy(synthetic.make_crops(B, seed)).to(device)
        emb, (mn, idx, sc) = rec(x)
        emb, mn, idx, sc = emb.clone(), mn.clone(), idx.clone(), sc.clone()
        want = hip_model.embed_tta(x) if tta else hip_model(x)
        wmn, widx, wsc = match_facebank(want, bank, 60)
        assert torch.equal(emb, want)
        assert torch.equal(idx, widx) and torch.equal(mn, wmn) and torch.equal(sc, wsc)


def test_graph_embeddings_only(hip_model, device):
    rec = GraphedRecognizer(hip_model, None)
    x = torch.from_numpy(synthetic.make_crops(2, 9)).to(device)
    emb, match = rec(x)
    assert match is None and torch.equal(emb, hip_model(x))
    # crops written into the graph's own input: no staging copy, same result
    buf = rec.input_buffer(2)
    y = torch.from_numpy(synthetic.make_crops(2, 10)).to(device)
    buf.copy_(y)
    emb2, _ = rec(buf)
    assert torch.equal(emb2, hip_model(y))
    with pytest.raises(RuntimeError):
        rec(torch.zeros((100, 3, 112, 112), device=device))
