"""Alignment warp on the device against the CPU oracle (bit exact: integer arithmetic on both sides).

The oracle itself restates OpenCV's warpAffine from its published algorithm and is NOT pinned against
cv2 (not installed here; the reference holds no aligned-crop fixtures): what these tests establish is
that the kernel equals the oracle, and that both have the properties any correct warp has (integer
translations copy pixels, the template maps onto itself, out-of-frame taps read 0).
"""
import numpy as np
import pytest
import torch

from oracle import alignment_ref as ar

pytestmark = pytest.mark.gpu


def _frame(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    base = (np.stack([xx * 3 + yy, xx + 2 * yy, 255 - xx - yy], -1) % 256).astype(np.int64)
    return ((base + rng.integers(0, 64, (H, W, 3))) % 256).astype(np.uint8)


def _landmarks(rng, n, H, W, scale_range=(0.5, 3.0)):
    T = np.asarray(ar.TEMPLATE_SQUARE)
    rows = []
    for _ in range(n):
        th = rng.uniform(-0.6, 0.6)
        s = rng.uniform(*scale_range)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        P = s * (T - 56.0) @ R.T + rng.uniform([0, 0], [W, H]) + rng.normal(0, 1.0, (5, 2))
        rows.append(np.concatenate([P[:, 0], P[:, 1]]))
    return np.asarray(rows)


@pytest.mark.parametrize("H,W,n", [(480, 640, 7), (1080, 1920, 16), (97, 131, 5), (112, 112, 3)])
def test_warp_equals_oracle(device, H, W, n):
    from face_recognition_amd import align_trans
    rng = np.random.default_rng(H * 7 + W)
    img = _frame(rng, H, W)
    lms = _landmarks(rng, n, H, W)                       # centres anywhere in the frame: many crops cross the border
    got = align_trans.align_faces(torch.from_numpy(img).to(device), lms).cpu().numpy()
    want = np.stack(ar.face_alignment(img, lms))
    assert got.shape == (n, 112, 112, 3) and got.dtype == np.uint8
    assert np.array_equal(got, want)
    assert (want == 0).all(-1).any() or H >= 480        # the small frames do exercise the zero border


def test_integer_translation_copies_pixels_and_border_is_zero(device):
    from face_recognition_amd import align_trans
    rng = np.random.default_rng(11)
    img = _frame(rng, 300, 400)
    d = torch.from_numpy(img).to(device)
    M = np.array([[1, 0, -37.0, 0, 1, -91.0],            # crop = img[91:203, 37:149]
                  [1, 0, 50.0, 0, 1, 20.0],              # shifted right/down: first 20 rows / 50 columns come from outside
                  [1, 0, -350.0, 0, 1, -250.0]])         # hangs over the bottom-right corner
    got = align_trans.warp_faces(d, M).cpu().numpy()
    assert np.array_equal(got[0], img[91:203, 37:149])
    assert (got[1][:20] == 0).all() and (got[1][:, :50] == 0).all() and np.array_equal(got[1][20:, 50:], img[:92, :62])
    assert np.array_equal(got[2][:50, :50], img[250:300, 350:400]) and (got[2][50:] == 0).all() and (got[2][:, 50:] == 0).all()
    for i in range(3):
        assert np.array_equal(got[i], ar.warp_affine(img, M[i].reshape(2, 3)))


def test_other_crop_size_zero_faces_and_errors(device):
    from face_recognition_amd import align_trans
    rng = np.random.default_rng(5)
    img = _frame(rng, 200, 200)
    d = torch.from_numpy(img).to(device)
    lms = _landmarks(rng, 4, 200, 200)
    M96 = align_trans.alignment_matrices(lms, align_trans.COORD5POINT_96)
    got = align_trans.warp_faces(d, M96, 112, 96).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], ar.warp_affine(img, M96[i].reshape(2, 3), 112, 96))
    assert align_trans.warp_faces(d, np.zeros((0, 6))).shape == (0, 112, 112, 3)
    assert align_trans.Face_alignment(img, True, []) == []
    with pytest.raises(RuntimeError, match="multiple of 4"):
        align_trans.warp_faces(d, M96, 112, 98)
    with pytest.raises(RuntimeError, match="uint8"):
        align_trans.warp_faces(d.float(), M96)
    with pytest.raises(NameError):
        align_trans.Face_alignment(img, False, lms)


def test_face_alignment_feeds_the_embedding_network(hip_model, device):
    """Frame -> crops -> embeddings without leaving the device equals the host round trip."""
    from face_recognition_amd import align_trans
    rng = np.random.default_rng(9)
    img = _frame(rng, 480, 640)
    lms = _landmarks(rng, 6, 480, 640, scale_range=(1.0, 2.0))
    crops_dev = align_trans.align_faces(torch.from_numpy(img).to(device), lms)
    crops_host = align_trans.Face_alignment(img, True, lms, device=device)
    assert len(crops_host) == 6 and np.array_equal(np.stack(crops_host), crops_dev.cpu().numpy())
    emb = hip_model.forward_u8(crops_dev, tta=True)
    emb2 = hip_model.forward_u8(torch.from_numpy(np.stack(ar.face_alignment(img, lms))).to(device), tta=True)
    assert torch.equal(emb, emb2)
    assert torch.allclose(emb.norm(dim=1), torch.ones(6, device=device), atol=1e-4)


def test_recognize_frame_against_the_oracle_chain(hip_model_calibrated, device, calibrated_state_dict_np):
    """One frame of the demos (cam_demo.py:79-105): aligned crops -> embeddings (with and without the mirrored copy) ->
    match, on the device in one call, against the oracle's alignment + forward + broadcast match on the host.  (Weights
    whose final BatchNorm statistics fit the activations: different faces are then far apart, as with trained weights.)"""
    hip_model, state_dict_np = hip_model_calibrated, calibrated_state_dict_np
    from face_recognition_amd.cam_demo import recognize_frame
    from oracle import matching_ref, mobilefacenet_ref
    rng = np.random.default_rng(21)
    img = _frame(rng, 360, 480)
    lms = _landmarks(rng, 5, 360, 480, scale_range=(1.0, 2.0))
    crops = np.stack(ar.face_alignment(img, lms))
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255).sub(0.5).div(0.5)
    embed = lambda t: mobilefacenet_ref.forward(state_dict_np, t)      # noqa: E731
    for tta in (False, True):
        want = embed(x)
        if tta:
            want = torch.nn.functional.normalize(want + embed(torch.flip(x, dims=[3])))
        # facebank: faces 3, 0 and 1 enrolled (in that order); faces 2 and 4 are nobody's.  Two thresholds: the demo's
        # default, and one so strict (d^2 <= 0.0125) that only a face's own enrolment passes it
        bank = want[[3, 0, 1]].clone()
        names = np.array(["", "carol", "alice", "bob"])
        for threshold in (60, 155):
            labels, score, results = recognize_frame(hip_model, img, lms, bank.to(device), names, threshold=threshold, tta=tta)
            omin, oidx, oscore = matching_ref.match_broadcast(want, bank, float(threshold))
            limit = (threshold - 156) / -80
            assert (omin - limit).abs().min() > 1e-3, "a distance too near the limit to pin the decision"
            assert results.cpu().tolist() == oidx.tolist()
            assert labels == [str(names[i + 1]) for i in oidx.tolist()]
            assert labels[0] == "alice" and labels[1] == "bob" and labels[3] == "carol"
            assert np.allclose(score.cpu().numpy(), oscore.numpy(), atol=0.1)
        assert labels[2] == "" and labels[4] == "" and results.cpu().tolist() == [1, 2, -1, 0, -1]
        # a frame already on the device
        labels2, _, _ = recognize_frame(hip_model, torch.from_numpy(img).to(device), lms, bank.to(device), names, threshold=155, tta=tta)
        assert labels2 == labels
    assert recognize_frame(hip_model, img, [], bank.to(device), names)[0] == []
