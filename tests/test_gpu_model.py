"""The whole MobileFaceNet forward on the GPU against the CPU oracle and the golden embeddings
recorded from the reference.  Bar (from the task statement): per-embedding cosine >= 0.9999."""
import os

import numpy as np
import pytest
import torch

from conftest import cosine_rows
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
from oracle import mobilefacenet_ref as ref

pytestmark = pytest.mark.gpu
COS_BAR = 0.9999


def test_golden_embeddings(hip_model, device, golden_dir):
    z = np.load(os.path.join(golden_dir, "embeddings_b4.npz"))
    x = torch.from_numpy(synthetic.make_crops(int(z["batch"]), int(z["input_seed"]))).to(device)
    y = hip_model(x).cpu().numpy()
    assert y.shape == (4, 512) and y.dtype == np.float32
    cos = cosine_rows(y, z["emb"])
    print("cosine vs reference:", cos)
    assert cos.min() >= COS_BAR
    assert np.allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-5)
    ys = hip_model(torch.from_numpy(synthetic.make_smooth_crops(4, int(z["input_seed"]))).to(device)).cpu().numpy()
    assert cosine_rows(ys, z["emb_smooth"]).min() >= COS_BAR
    yt = hip_model.embed_tta(x).cpu().numpy()
    assert cosine_rows(yt, z["emb_tta"]).min() >= COS_BAR


def test_layer_by_layer_against_oracle(hip_model, device, state_dict_np):
    """Every kernel's output activation vs the oracle's tap for the same layer (relative L2 error)."""
    B = 2
    xn = synthetic.make_crops(B, 3)
    taps = {}
    ref.forward(state_dict_np, torch.from_numpy(xn), taps)
    x = torch.from_numpy(xn).to(device)
    hip_model(x)  # builds the handle
    table = hip_model.op_table()
    worst = 0.0
    for i, (name, kind, *_rest) in enumerate(table[:-1]):
        # oracle taps exist at block granularity: stem, conv2_dw, block outputs (= project ops), conv_6_sep, conv_6_dw
        key = name[:-len(".project")] if name.endswith(".project") else name
        if key not in taps or name.endswith(".conv") or name.endswith(".conv_dw"):
            continue
        act = hip_model.debug_prefix(x, i + 1).cpu()
        t = taps[key].permute(0, 2, 3, 1).reshape(B, -1)
        rel = float((act - t).norm() / t.norm())
        worst = max(worst, rel)
        assert rel < 4e-3, f"op {i} {name}: relative error {rel:.3g}"
    print("worst relative activation error", worst)


@pytest.mark.parametrize("B", [1, 2, 5, 33])
def test_batch_sizes_match_oracle(hip_model, device, state_dict_np, B):
    xn = synthetic.make_crops(B, 100 + B)
    want = ref.forward_numpy(state_dict_np, xn)
    got = hip_model(torch.from_numpy(xn).to(device)).cpu().numpy()
    assert cosine_rows(got, want).min() >= COS_BAR


def test_batch_independence_and_chunking(hip_model, device):
    """Row i of a batched call equals the B=1 call on crop i; slicing the batch changes nothing."""
    x = torch.from_numpy(synthetic.make_crops(9, 7)).to(device)
    full = hip_model(x).cpu()
    one = torch.cat([hip_model(x[i:i + 1]).cpu() for i in range(9)])
    assert torch.equal(full, one)
    hip_model.set_batch_chunk(4)
    try:
        chunked = hip_model(x).cpu()
    finally:
        hip_model.set_batch_chunk(0)
    assert torch.equal(full, chunked)


def test_fused_and_unfused_schedules_agree(hip_model, device, state_dict_np):
    """One kernel per block vs one kernel per conv: same arithmetic up to summation order."""
    xn = synthetic.make_crops(3, 21)
    x = torch.from_numpy(xn).to(device)
    fused = hip_model(x).cpu().numpy()
    hip_model.set_fused(False)
    try:
        plain = hip_model(x).cpu().numpy()
        assert len(hip_model.op_table()) == 50
    finally:
        hip_model.set_fused(True)
    assert len(hip_model.op_table()) == 19
    want = ref.forward_numpy(state_dict_np, xn)
    assert cosine_rows(fused, want).min() >= COS_BAR and cosine_rows(plain, want).min() >= COS_BAR
    assert cosine_rows(fused, plain).min() >= 0.99999


def test_output_is_fresh_tensor_and_deterministic(hip_model, device):
    x = torch.from_numpy(synthetic.make_crops(3, 8)).to(device)
    a = hip_model(x)
    b = hip_model(x)
    assert a.data_ptr() != b.data_ptr() and torch.equal(a, b)
    assert not a.requires_grad


def test_bf16_variant(device, state_dict_np):
    """bf16 storage keeps 8 significant bits per activation; the bar is looser and stated here."""
    m = MobileFaceNet(512, act_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()})
    m = m.to(device).eval()
    xn = synthetic.make_crops(4, 0)
    want = ref.forward_numpy(state_dict_np, xn)
    got = m(torch.from_numpy(xn).to(device)).cpu().numpy()
    cos = cosine_rows(got, want)
    print("bf16 cosine:", cos)
    assert cos.min() >= 0.999


def test_embedding_size_128(device, golden_dir):
    z = np.load(os.path.join(golden_dir, "embeddings_b4.npz"))
    sd = synthetic.make_state_dict(7, 128)
    m = MobileFaceNet(128)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(device).eval()
    xn = synthetic.make_crops(4, int(z["input_seed"]))[:2]
    got = m(torch.from_numpy(xn).to(device)).cpu().numpy()
    assert got.shape == (2, 128)
    assert cosine_rows(got, z["emb128_seed7_b2"]).min() >= COS_BAR


def test_reloading_weights_changes_output(device, state_dict_np):
    m = MobileFaceNet(512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()})
    m = m.to(device).eval()
    x = torch.from_numpy(synthetic.make_crops(2, 1)).to(device)
    a = m(x).cpu()
    sd2 = synthetic.make_state_dict(2, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd2.items()})
    b = m(x).cpu().numpy()
    want = ref.forward_numpy(sd2, x.cpu().numpy())
    assert cosine_rows(b, want).min() >= COS_BAR
    assert not np.allclose(a.numpy(), b)


def test_errors(hip_model, device):
    with pytest.raises(RuntimeError):
        hip_model(torch.zeros((1, 3, 112, 112)))            # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        hip_model(torch.zeros((1, 3, 64, 64), device=device))
    hip_model.train()
    try:
        with pytest.raises(RuntimeError):
            hip_model(torch.zeros((1, 3, 112, 112), device=device))
    finally:
        hip_model.eval()
