"""The CPU oracle against the golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import os

import numpy as np
import torch

from face_recognition_amd import synthetic
from oracle import evaluation_ref, matching_ref, mobilefacenet_ref

EVAL_CASES = (("p600", 600, 5, {}), ("p1200_hard", 1200, 11, dict(same_cos=0.4, diff_cos=0.2)))


def test_oracle_embeddings_match_reference_golden(golden_dir, state_dict_np):
    z = np.load(os.path.join(golden_dir, "embeddings_b4.npz"))
    assert int(z["weight_seed"]) == 1
    x = synthetic.make_crops(int(z["batch"]), int(z["input_seed"]))
    y = mobilefacenet_ref.forward_numpy(state_dict_np, x)
    assert np.abs(y - z["emb"]).max() < 5e-6            # same torch build: bit-equal; other builds: fp32 round-off
    yt = mobilefacenet_ref.embed_tta(state_dict_np, torch.from_numpy(x)).numpy()
    assert np.abs(yt - z["emb_tta"]).max() < 5e-6
    ys = mobilefacenet_ref.forward_numpy(state_dict_np, synthetic.make_smooth_crops(4, int(z["input_seed"])))
    assert np.abs(ys - z["emb_smooth"]).max() < 5e-6
    y128 = mobilefacenet_ref.forward_numpy(synthetic.make_state_dict(7, 128), x[:2])
    assert y128.shape == (2, 128) and np.abs(y128 - z["emb128_seed7_b2"]).max() < 5e-6


def test_oracle_matches_reference_on_calibrated_weights(golden_dir, state_dict_np):
    """Weights whose final BatchNorm carries the statistics of the `linear` layer's output (measured with the reference
    model, stored in the fixture): embeddings spread over the sphere; the oracle reproduces the reference's outputs."""
    import os
    z = np.load(os.path.join(golden_dir, "embeddings_calibrated.npz"))
    sd = synthetic.with_embedding_statistics(state_dict_np, z["bn_running_mean"], z["bn_running_var"])
    left, right, _ = synthetic.pair_crops(int(z["crop_pairs"]), seed=int(z["crop_seed"]), noise=float(z["crop_noise"]))
    n = int(z["crops_used_per_side"])
    x = torch.from_numpy(np.concatenate([left[:n], right[:n]]))
    y = mobilefacenet_ref.forward(sd, x).numpy()
    assert np.abs(y - z["emb"]).max() < 2e-6
    assert np.abs(mobilefacenet_ref.embed_tta(sd, x).numpy() - z["emb_tta"]).max() < 2e-6
    c = z["emb"] @ z["emb"].T
    assert abs(c[~np.eye(len(c), dtype=bool)].mean()) < 0.2          # spread, not clustered


def test_oracle_rows_are_unit_norm_and_flip_matters(state_dict_np):
    x = torch.from_numpy(synthetic.make_crops(2, 5))
    y = mobilefacenet_ref.forward(state_dict_np, x)
    assert torch.allclose(y.norm(dim=1), torch.ones(2), atol=1e-6)
    yf = mobilefacenet_ref.forward(state_dict_np, torch.flip(x, dims=[3]))
    assert not torch.allclose(y, yf, atol=1e-3)


def test_oracle_l2norm_zero_row_nan():
    x = torch.zeros((2, 8))
    x[1] = 1
    y = mobilefacenet_ref.l2_norm(x)
    assert torch.isnan(y[0]).all() and torch.allclose(y[1].norm(), torch.tensor(1.0))


def test_oracle_evaluation_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "eval10fold.npz"))
    for tag, pairs, seed, kw in EVAL_CASES:
        assert int(z[f"{tag}_seed"]) == seed
        fl, fr, ds = synthetic.planted_features(pairs, seed=seed, **kw)
        for method in ("l2_distance", "cos_distance"):
            accs, thr = evaluation_ref.evaluation_10_fold(fl, fr, ds.folds, ds.flags, method)
            assert np.array_equal(accs, z[f"{tag}_{method}_accs"]), (tag, method)
            assert np.array_equal(thr, z[f"{tag}_{method}_thr"]), (tag, method)


def test_oracle_threshold_edge_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "eval10fold.npz"))
    grid = np.arange(-50, 51) * 3.0 / 50
    for method in ("l2_distance", "cos_distance"):
        assert evaluation_ref.get_threshold(z["edge_scores"], z["edge_flags"], 50, method) == float(z[f"edge_{method}_thr"])
        curve = np.array([evaluation_ref.get_accuracy(z["edge_scores"], z["edge_flags"], g, method) for g in grid])
        assert np.array_equal(curve, z[f"edge_{method}_acc_curve"])


def test_oracle_match_on_shipped_facebank(golden_dir):
    z = np.load(os.path.join(golden_dir, "facebank_match.npz"))
    fb, q = torch.from_numpy(z["facebank"]), torch.from_numpy(z["queries"])
    assert list(z["names"]) == ["", "Howard", "Leonard", "Penny", "Raj", "Sheldon"]
    for th in (60, 50):
        mn, idx, sc = matching_ref.match_broadcast(q, fb, float(th))
        assert np.array_equal(idx.numpy(), z[f"idx_th{th}"])
        assert np.allclose(mn.numpy(), z[f"min_th{th}"], atol=1e-6)
        assert np.allclose(sc.numpy(), z[f"score_th{th}"], atol=1e-4)
    # known answers of the shipped file: self-match, every cross pair rejected at the default threshold
    mn, idx, _ = matching_ref.match_broadcast(fb, fb, 60.0)
    assert idx.tolist() == [0, 1, 2, 3, 4] and float(mn.abs().max()) == 0.0
    d2 = z["pairwise_d2"]
    off = d2[~np.eye(5, dtype=bool)]
    assert abs(off.min() - 1.8266) < 1e-3 and abs(off.max() - 2.2806) < 1e-3
    assert (z["idx_th60"][10:15] == [-1, -1, 2, -1, 4]).all() and (z["idx_th50"][10:15] == [0, 1, 2, 3, 4]).all()


def test_oracle_bruteforce_and_merge_agree():
    g = synthetic.unit_rows(300, 64, seed=1)
    g[17] = g[5]                                  # exact tie -> lowest index first
    q = np.concatenate([g[[5]], synthetic.unit_rows(6, 64, seed=2)])
    d, i = matching_ref.l2_topk_bruteforce(q, g, 4, block=64)
    assert i[0, 0] == 5 and i[0, 1] == 17
    full = ((q[:, None, :].astype(np.float64) - g[None].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(np.sort(full, 1)[:, :4], d, atol=1e-9)
    parts_d, parts_i = [], []
    for b, e in ((0, 100), (100, 200), (200, 300)):
        pd, pi = matching_ref.l2_topk_bruteforce(q, g[b:e], 4)
        parts_d.append(pd)
        parts_i.append(pi + b)
    md, mi = matching_ref.merge_topk(np.stack(parts_d), np.stack(parts_i), 4)
    assert np.array_equal(mi, i) and np.allclose(md, d)
