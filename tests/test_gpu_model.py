"""The whole MobileFaceNet forward on the GPU against the CPU oracle and the golden embeddings
recorded from the reference.  Bar (from the task statement): per-embedding cosine >= 0.9999."""
import os

import numpy as np
import pytest
import torch

from conftest import cosine_rows
from face_recognition_amd import _lib, synthetic
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


def _calibrated_golden(golden_dir, state_dict_np):
    z = np.load(os.path.join(golden_dir, "embeddings_calibrated.npz"))
    sd = synthetic.with_embedding_statistics(state_dict_np, z["bn_running_mean"], z["bn_running_var"])
    left, right, _ = synthetic.pair_crops(int(z["crop_pairs"]), seed=int(z["crop_seed"]), noise=float(z["crop_noise"]))
    n = int(z["crops_used_per_side"])
    return z, sd, np.concatenate([left[:n], right[:n]])


@pytest.mark.parametrize("dt,bar", [("f16", COS_BAR), ("bf16", 0.998)])
def test_golden_embeddings_of_calibrated_weights(device, golden_dir, state_dict_np, dt, bar):
    """The bar where it is honest.  The seeded weights' embeddings cluster around one direction (pairwise cosine 0.8),
    which flatters a per-embedding cosine tenfold; with the final BatchNorm's statistics set to what the `linear`
    layer really outputs -- measured with the REFERENCE model when the fixture was made, as a trained network has them --
    the 16 embeddings spread over the sphere (pairwise cosine 0.06 on average) and the figure means what it says.
    Expected values are the reference's own outputs (tests/golden/make_golden.py).  f16 activations meet the task's bar
    of 0.9999; bf16 activations (8 significant bits) do not and are held to what they measure."""
    z, sd, crops = _calibrated_golden(golden_dir, state_dict_np)
    m = MobileFaceNet(512, act_dtype=dt)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(device).eval()
    x = torch.from_numpy(crops).to(device)
    cos = cosine_rows(m(x).cpu().numpy(), z["emb"])
    cos_tta = cosine_rows(m.embed_tta(x).cpu().numpy(), z["emb_tta"])
    print(dt, "cosine vs the reference, calibrated weights: min %.6f, tta min %.6f" % (cos.min(), cos_tta.min()))
    assert cos.min() >= bar and cos_tta.min() >= bar


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


def test_large_batch_rows_match_small_batches(hip_model, device, invariant_bits):
    """A batch large enough that every workgroup of the block kernels walks over several images
    gives, row for row, what small batches give."""
    B = 700
    x = torch.from_numpy(synthetic.make_crops(64, 11)).to(device).repeat(11, 1, 1, 1)[:B].contiguous()
    x += torch.linspace(-0.2, 0.2, B, device=device).view(B, 1, 1, 1)
    full = hip_model(x).cpu()
    for lo in (0, 255, 256, 511, 693):
        part = hip_model(x[lo:lo + 7]).cpu()
        assert torch.equal(full[lo:lo + 7], part), lo
    assert torch.isfinite(full).all()


def test_batch_1024_rows_match_small_batches(hip_model, device, invariant_bits):
    """The benchmark's batch (1024 crops): the block kernels split it into groups of images per resident
    workgroup differently from any smaller batch; rows must not depend on that."""
    B = 1024
    x = torch.from_numpy(synthetic.make_crops(64, 17)).to(device).repeat(16, 1, 1, 1).contiguous()
    x += torch.linspace(-0.25, 0.25, B, device=device).view(B, 1, 1, 1)
    full = hip_model(x).cpu()
    assert torch.isfinite(full).all()
    for lo in (0, 3, 255, 256, 509, 767, 1017):
        part = hip_model(x[lo:lo + 7]).cpu()
        assert torch.equal(full[lo:lo + 7], part), lo
    mid = hip_model(x[300:500].contiguous()).cpu()          # 200 crops: the reference harness's 4 x bs 50
    assert torch.equal(full[300:500], mid)


def test_rows_do_not_depend_on_the_batch_at_every_launcher_limit(hip_model, device, invariant_bits):
    """Batch sizes on both sides of every limit at which the library changes kernels or workgroup shapes (short / tall
    bands at 18 and 36, conv_5's bands / two / four stacked images at 36 and 512, the conv_4 chain at 36, conv_3's
    pipeline ranges 193-256, 385-512, 641-, a second image per workgroup above 256 / 1024, batches that are no multiple
    of a stack): every face's embedding is the bits it has in the largest batch."""
    B = 1025
    x = torch.from_numpy(synthetic.make_crops(41, 29)).to(device).repeat(25, 1, 1, 1).contiguous()
    x += torch.linspace(-0.3, 0.3, B, device=device).view(B, 1, 1, 1)
    full = hip_model(x).cpu()
    assert torch.isfinite(full).all()
    for b in (1, 2, 3, 18, 19, 36, 37, 38, 39, 192, 193, 256, 257, 384, 385, 511, 512, 513, 514, 640, 641, 1023, 1024):
        assert torch.equal(hip_model(x[:b].contiguous()).cpu(), full[:b]), b
        assert torch.equal(hip_model(x[B - b:].contiguous()).cpu(), full[B - b:]), -b


def test_chained_and_per_block_launches_agree_bit_for_bit(hip_model, device, invariant_bits):
    """conv_4's six blocks as one register-resident launch (large batches) vs six launches: same rows; and the
    launch plan says which of the two a batch size gets."""
    x = torch.from_numpy(synthetic.make_crops(150, 19)).to(device)
    want = hip_model(x).cpu()
    plan = hip_model.launch_plan(150)
    names = [r[0] for r in hip_model.op_table()]
    first = names.index("conv_4.model.0")
    assert plan[first:first + 6] == [first] * 6                       # one launch covers the six blocks
    # small batch: conv_4 as three launches of two blocks each (two-row bands that compute their neighbours' nearest
    # rows themselves), conv_5 as one, every other op a launch of its own
    small = list(range(len(names)))
    for j in range(6):
        small[first + j] = first + 2 * (j // 2)
    five = names.index("conv_5.model.0")
    small[five + 1] = five                                            # conv_5's two blocks: one launch of one-row bands
    assert hip_model.launch_plan(8) == small
    hip_model.set_chain(False)
    try:
        assert hip_model.launch_plan(150) == list(range(len(names))) == hip_model.launch_plan(8)
        assert torch.equal(hip_model(x).cpu(), want)                                 # whole images, one block per launch
        assert torch.equal(hip_model(x[:60].contiguous()).cpu(), want[:60])          # 7-row bands
        assert torch.equal(hip_model(x[:8].contiguous()).cpu(), want[:8])            # 2-row bands, one block per launch
    finally:
        hip_model.set_chain(True)
    assert torch.equal(hip_model(x[:8].contiguous()).cpu(), want[:8])                # 2-row bands, two blocks per launch
    assert torch.equal(hip_model(x[:36].contiguous()).cpu(), want[:36])
    assert torch.equal(hip_model(x[:1].contiguous()).cpu(), want[:1])
    assert hip_model.launch_plan(60)[first:first + 6] == [first] * 6
    assert torch.equal(hip_model(x[:60].contiguous()).cpu(), want[:60])              # one launch on a part-filled chip


def test_row_pipeline_kernel_matches_band_kernels(device, hip_model, state_dict_np, invariant_bits):
    """conv_3 runs as a pipeline inside a workgroup (producer / consumer waves hand tiles of 32 consecutive pixels on
    through LDS, flags instead of workgroup-wide barriers) where the band kernels' last round over the chip would be
    part empty, as band kernels elsewhere; both give the same bits.  512 crops: pipeline, two images per workgroup
    (49 whole tiles); 256: pipeline, one image (its last tile half empty); 300 and 7: bands.  The same with bf16
    activations."""
    B = 512
    x = torch.from_numpy(synthetic.make_crops(64, 23)).to(device).repeat(8, 1, 1, 1).contiguous()
    x += torch.linspace(-0.25, 0.25, B, device=device).view(B, 1, 1, 1)
    mb = MobileFaceNet(512, act_dtype="bf16")
    mb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()})
    mb = mb.to(device).eval()
    if torch.cuda.get_device_properties(device).multi_processor_count == 256:
        assert [2 in hip_model.op_variants(b) for b in (512, 256, 300, 7)] == [True, True, False, False]
    for model in (hip_model, mb):
        full = model(x).cpu()
        assert torch.isfinite(full).all()
        assert torch.equal(model(x[:256].contiguous()).cpu(), full[:256])
        assert torch.equal(model(x[:300].contiguous()).cpu(), full[:300])
        for lo in (0, 255, 256, 505):
            assert torch.equal(model(x[lo:lo + 7].contiguous()).cpu(), full[lo:lo + 7]), lo


def test_band_configurations_agree_bit_for_bit(hip_model, device, invariant_bits):
    """The block kernels run short bands while those fit the chip in one go (up to 18 or 36 images, depending on
    the layer) and tall bands beyond; conv_3 has two tall heights (7 and 14 rows, whichever needs less time for its
    rounds over the chip: 14 rows at 100 images, 7 rows at 40 and 64).  Batches on every side of the limits give
    the same rows."""
    x = torch.from_numpy(synthetic.make_crops(100, 13)).to(device)
    want = hip_model(x).cpu()                      # 100: tall bands everywhere, 14-row bands in conv_3
    for B in (1, 18, 19, 36, 37, 40, 64):
        assert torch.equal(hip_model(x[:B].contiguous()).cpu(), want[:B]), B


def test_small_batch_mode(hip_model_calibrated, device, calibrated_state_dict_np):
    """The default mode runs the band launches of small batches (7 B <= compute units: the per-frame loop of
    cam_demo.py:87-105) with two slices of a block side by side and adds the block's project sum in two halves.
    Against the fp32 oracle on calibrated weights: cosine >= 0.9999 for every batch size from 1 to 8; inside the regime
    a face's embedding is the same bits whatever the batch; against the batch-invariant mode (and so against large
    batches) it agrees to rounding; above the limit the mode changes nothing."""
    from face_recognition_amd import face_model
    model = hip_model_calibrated
    assert face_model.small_batch_mode() == 1 and not model.batch_invariant
    limit = model.batch_invariant_from(device) - 1
    assert limit == torch.cuda.get_device_properties(device).multi_processor_count // 7
    xn = synthetic.make_crops(limit + 4, 31)
    x = torch.from_numpy(xn).to(device)
    want = ref.forward_numpy(calibrated_state_dict_np, xn[:8])
    fast = model(x[:limit].contiguous()).cpu()
    for B in range(1, 9):
        got = model(x[:B].contiguous()).cpu()
        assert cosine_rows(got.numpy(), want[:B]).min() >= COS_BAR, B
        assert torch.equal(got, fast[:B]), B
    assert torch.equal(model(x[limit - 1:limit].contiguous()).cpu(), fast[limit - 1:])
    above = model(x).cpu()
    face_model.set_small_batch_mode(0)
    try:
        assert model.batch_invariant and model.batch_invariant_from(device) == 1
        exact = model(x[:limit].contiguous()).cpu()
        assert torch.equal(model(x).cpu(), above) and torch.equal(exact, above[:limit])
    finally:
        face_model.set_small_batch_mode(1)
    assert not torch.equal(fast, exact)
    assert cosine_rows(fast.numpy(), exact.numpy()).min() >= 0.99999          # measured 0.999996: one rounding of a stored activation
    with pytest.raises(RuntimeError, match="mode"):
        face_model.set_small_batch_mode(2)


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
    assert len(hip_model.op_table()) == 18
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


def test_a_wait_that_gives_up_is_reported(hip_model, device):
    """The kernels that synchronise their waves through flags in LDS (the conv_3 tile pipeline, the conv_4 / conv_5
    chains) bound their polls.  With the bound set to zero every wait gives up at once: the embeddings are garbage, and
    the library must say so -- the next forward on the device fails with the kernel's name instead of returning them
    with status 0 -- and work normally again afterwards."""
    import ctypes
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((256, 3, 112, 112), generator=g) * 2 - 1).to(device)   # tile pipeline and both chains
    good = hip_model(x)
    torch.cuda.synchronize()
    assert lib.fra_device_fault(device.index or 0, None) == 0
    try:
        lib.fra_dbg_set_spin_limit(0)
        hip_model(x)                                       # rc 0: the launches were queued; their waits give up on the device
        torch.cuda.synchronize()
    finally:
        lib.fra_dbg_set_spin_limit(-1)
    with pytest.raises(RuntimeError, match="gave up"):
        hip_model(x)
    tag = ctypes.c_uint(0)
    assert lib.fra_device_fault(device.index or 0, ctypes.byref(tag)) == 0 and tag.value == 0   # reported once, then cleared
    again = hip_model(x)
    torch.cuda.synchronize()
    assert torch.equal(again, good)
    assert lib.fra_device_fault(device.index or 0, None) == 0


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


def _u8_crops(n, seed):
    rng = np.random.default_rng(seed)
    coarse = rng.integers(0, 256, (n, 14, 14, 3), dtype=np.uint8)
    img = np.repeat(np.repeat(coarse, 8, 1), 8, 2).astype(np.int16) + rng.integers(-20, 21, (n, 112, 112, 3))
    return np.clip(img, 0, 255).astype(np.uint8)            # HWC, BGR like cv2.imread


def _transform(u8):
    """ToTensor + Normalize(0.5, 0.5) of facebank.py:22-24 on a uint8 HWC batch -> f32 NCHW."""
    t = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


@pytest.mark.parametrize("B", [1, 6])
def test_uint8_input_and_fused_tta(hip_model, device, state_dict_np, B):
    """Raw uint8 HWC crops with the transform fused into the first kernel, and the flip-TTA with the
    mirrored read fused as well, against the oracle on the explicitly transformed / flipped tensors."""
    u8 = _u8_crops(B, 40 + B)
    x = _transform(u8)
    want = ref.forward(state_dict_np, x).numpy()
    want_tta = ref.embed_tta(state_dict_np, x).numpy()
    xd = torch.from_numpy(u8).to(device)
    got = hip_model.forward_u8(xd).cpu().numpy()
    got_tta = hip_model.forward_u8(xd, tta=True).cpu().numpy()
    assert cosine_rows(got, want).min() >= COS_BAR
    assert cosine_rows(got_tta, want_tta).min() >= COS_BAR
    # the f32 entry point on the transformed tensor gives the same embeddings as the uint8 entry point
    same = hip_model(x.to(device)).cpu().numpy()
    assert np.abs(same - got).max() < 2e-3 and cosine_rows(same, got).min() >= 0.99999
    # fused flip == explicit flip through the plain forward
    flipped = hip_model(torch.flip(x, dims=[3]).to(device)).cpu()
    manual = ((torch.from_numpy(same) + flipped) / (torch.from_numpy(same) + flipped).norm(dim=1, keepdim=True)).numpy()
    assert cosine_rows(hip_model.embed_tta(x.to(device)).cpu().numpy(), manual).min() >= 0.99999
    with pytest.raises(RuntimeError):
        hip_model.forward_u8(xd.float())


def test_prepare_facebank_from_image_folders(hip_model, device, state_dict_np, tmp_path):
    """Enrolment (facebank.py:33-77): per-identity mean of TTA embeddings, not re-normalised; names[0] == '';
    files written in the reference's format and readable by load_facebank."""
    from PIL import Image

    from face_recognition_amd.facebank import load_facebank, match_facebank, prepare_facebank
    people = {"alice": _u8_crops(3, 1), "bob": _u8_crops(1, 2), "carol": _u8_crops(2, 3)}
    for name, crops in people.items():
        (tmp_path / name).mkdir()
        for i, c in enumerate(crops):
            Image.fromarray(c[:, :, ::-1]).save(tmp_path / name / f"{i}.png")      # file is RGB, reader returns BGR
    (tmp_path / "notes.txt").write_text("files at the top level are skipped")
    (tmp_path / "alice" / ".hidden").write_text("hidden files are skipped")
    os.remove(tmp_path / "alice" / ".hidden")
    emb, names = prepare_facebank(hip_model, path=str(tmp_path), tta=True)
    assert names[0] == "" and sorted(names[1:]) == sorted(people) and emb.shape == (3, 512)
    for row, name in zip(emb.cpu().numpy(), names[1:]):
        want = ref.embed_tta(state_dict_np, _transform(people[name])).mean(0).numpy()
        assert cosine_rows(row[None], want[None]).min() >= COS_BAR
        assert abs(np.linalg.norm(row) - np.linalg.norm(want)) < 2e-3            # mean of unit rows: norm <= 1
    emb2, names2 = load_facebank(str(tmp_path))
    assert torch.equal(emb2.cpu(), emb.cpu()) and list(names2) == list(names)
    q = hip_model.forward_u8(torch.from_numpy(people["bob"]).to(device), tta=True)
    mn, idx, _ = match_facebank(q, emb, threshold=60)
    assert names[int(idx[0]) + 1] == "bob" and float(mn[0]) < 1e-3


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the contract's keys (a short run at the benchmark's batch size, a small
    gallery, no CPU leg)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1024",
                        "--gallery-rows", "2200000", "--queries", "256", "--search-steps", "1", "--no-cpu", "--no-latency"],
                       capture_output=True, text=True, timeout=280, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "faces/s" and r["value"] > 0
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and "workload" in r["config"]
    assert "traffic_source" in r["roofline"] and "traffic" in r["roofline"]
    by_class = r["roofline_by_class"]
    assert set(by_class) == {"depthwise_3x3_and_gdconv", "pointwise_1x1_and_linear", "stem"}
    for cls in by_class.values():
        assert cls["bound"] in ("hbm", "mfma") and abs(cls["frac"] - cls["achieved"] / cls["peak"]) < 1e-3
    assert by_class["pointwise_1x1_and_linear"]["bound"] == "mfma" and by_class["depthwise_3x3_and_gdconv"]["bound"] == "hbm"
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == ("GB/s" if rf["bound"] == "hbm" else "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the in-CU floor beside the two rooflines: per kernel the clocks its executed instructions need on the matrix pipe,
    # the LDS array and the vector ALU (committed profile of this batch size), against this run's launch time
    assert r["incu_source"] and r["incu_source"].startswith("profiles/")
    for name, k in r["kernels"].items():
        u = k["incu"]
        assert u is not None, name
        assert u["bound"] in ("mfma", "lds", "valu") and u["floor_us"] == max(u["mfma_us"], u["lds_us"], u["valu_us"])
        assert 0 < u["frac"] <= 1.0 and abs(u["frac"] - u["floor_us"] / u["launch_us"]) < 2e-3, (name, u)
        assert u["above_hbm_floor"] == (u["floor_us"] > u["hbm_floor_us"])
    assert rf["incu"] == r["kernels"][rf["kernel"]]["incu"]
    # the sharded search: a digest to compare runs at different world sizes with, the three steps apart, and one rank's
    # share of an 8-GPU run rehearsed on this GPU
    for s in (r["search_1m"], r["search"]):
        assert len(s["result_digest"]) == 16 and set(s["shard"]) >= {"rows_per_gpu", "local_search_ms", "allgather_ms", "merge_ms"}
    s = r["search"]
    assert s["shard"]["rows_per_gpu"] == 2200000 and s["predicted_8gpu"]["rows_per_gpu"] == 275000
    assert abs(s["predicted_8gpu_ms"] - (s["predicted_8gpu"]["local_search_ms"] + s["predicted_8gpu"]["merge_ms"]
                                         + s["predicted_8gpu"]["allgather_assumed_ms"])) < 2e-3
