"""Verification protocol end to end on the GPU: getFeature on synthetic pair images against the
oracle embeddings, and 10-fold accuracy within 0.1 pt of the fp32 CPU path (the task's bar)."""
import numpy as np
import pytest
import torch

import os

from conftest import cosine_rows
from face_recognition_amd import Evaluation, synthetic
from oracle import evaluation_ref, mobilefacenet_ref

pytestmark = pytest.mark.gpu


def _loader(left, right, bs):
    L, R = torch.from_numpy(left), torch.from_numpy(right)
    for i in range(0, L.shape[0], bs):
        l, r = L[i:i + bs], R[i:i + bs]
        yield [l, torch.flip(l, dims=[3]), r, torch.flip(r, dims=[3])]     # cv2.flip(img, 1) == flip along W


def test_get_feature_and_10fold_vs_oracle(hip_model, device, state_dict_np):
    pairs = 200
    left, right, ds = synthetic.pair_crops(pairs, seed=6, noise=0.3)
    fl, fr = Evaluation.getFeature(hip_model, _loader(left, right, 50), device, flip=True)
    assert fl.device.type == "cpu" and fl.shape == (pairs, 512) and fl.dtype == torch.float32
    embed = lambda x: mobilefacenet_ref.forward(state_dict_np, x)      # noqa: E731
    ol, orr = evaluation_ref.get_feature(embed, _loader(left, right, 50), flip=True)
    assert cosine_rows(fl.numpy(), ol.numpy()).min() >= 0.9999
    assert cosine_rows(fr.numpy(), orr.numpy()).min() >= 0.9999
    for method in ("l2_distance", "cos_distance"):
        acc, thr = Evaluation.evaluation_10_fold(fl, fr, ds, method)
        oacc, othr = evaluation_ref.evaluation_10_fold(ol.numpy(), orr.numpy(), ds.folds, ds.flags, method)
        # the protocol itself is exact: the oracle's 10-fold on the GPU features gives the same numbers
        pacc, pthr = evaluation_ref.evaluation_10_fold(fl.numpy(), fr.numpy(), ds.folds, ds.flags, method)
        assert np.array_equal(acc, pacc) and np.array_equal(thr, pthr)
        print(method, "acc %.3f vs oracle %.3f; thr %.4f vs %.4f" % (acc.mean() * 100, oacc.mean() * 100, thr.mean(), othr.mean()))
        # 200 pairs at ~99 % accuracy (the regime of the real benchmarks): one pair is 0.5 pt, and with
        # embeddings at cosine >= 0.9999 at most one borderline pair may change sides
        assert abs(acc.mean() - oacc.mean()) * 100 <= 0.5 + 1e-9
        assert abs(thr.mean() - othr.mean()) < 5e-3
    fl2, _ = Evaluation.getFeature(hip_model, _loader(left[:20], right[:20], 7), device, flip=False)
    want = mobilefacenet_ref.forward(state_dict_np, torch.from_numpy(left[:20])).numpy()
    assert cosine_rows(fl2.numpy(), want).min() >= 0.9999
    # getFeature groups loader batches into forwards of up to 1024 crops (the default for this network); a forward
    # per loader batch, as the reference runs them, gives the same bits
    assert hip_model.batch_invariant or hip_model.batch_invariant_from(device) <= 200    # 50 pairs x 4 crops per loader batch
    sl, sr = Evaluation.getFeature(hip_model, _loader(left, right, 50), device, flip=True, group_crops=0)
    assert torch.equal(sl, fl) and torch.equal(sr, fr)


def test_accuracy_within_point_one_on_gpu_embeddings(hip_model_calibrated, device, calibrated_state_dict_np):
    """The 0.1-pt criterion on REAL GPU embeddings at a size where it can fail: 2,000 synthetic pairs
    (one pair = 0.05 pt; fold = i // (P/10) as the LFW / CFP-FP / AgeDB loaders assign it) whose noise
    level puts the 10-fold accuracy in the low 90s, where many pairs sit near the threshold.  Embedded
    once by the HIP forward through getFeature (bs 50, flip) and once by the fp32 oracle on the host.  The weights
    carry final-BatchNorm statistics that match the network's own activations, as trained weights do: with arbitrary
    statistics all embeddings share one direction, the protocol's mean-centring removes it and magnifies storage
    rounding tenfold, which says nothing about a trained network."""
    pairs = 2000
    left, right, ds = synthetic.pair_crops(pairs, seed=16, noise=0.45)
    fl, fr = Evaluation.getFeature(hip_model_calibrated, _loader(left, right, 50), device, flip=True)
    embed = lambda x: mobilefacenet_ref.forward(calibrated_state_dict_np, x)      # noqa: E731
    ol, orr = evaluation_ref.get_feature(embed, _loader(left, right, 250), flip=True)
    assert cosine_rows(fl.numpy(), ol.numpy()).min() >= 0.9999
    assert cosine_rows(fr.numpy(), orr.numpy()).min() >= 0.9999
    for method in ("l2_distance", "cos_distance"):
        acc, thr = Evaluation.evaluation_10_fold(fl, fr, ds, method)
        oacc, othr = evaluation_ref.evaluation_10_fold(ol.numpy(), orr.numpy(), ds.folds, ds.flags, method)
        print(method, "GPU acc %.3f %% vs oracle %.3f %%; thr %.4f vs %.4f" %
              (acc.mean() * 100, oacc.mean() * 100, thr.mean(), othr.mean()))
        at_ref, _ = evaluation_ref.evaluation_10_fold(fl.numpy(), fr.numpy(), ds.folds, ds.flags, method,
                                                      fixed_thresholds=othr)
        print("   GPU features at the oracle's thresholds: %.3f %%" % (at_ref.mean() * 100))
        assert 70.0 < oacc.mean() * 100 < 99.5, "the set must leave pairs near the threshold"
        assert abs(at_ref.mean() - oacc.mean()) * 100 <= 0.1 + 1e-9      # accuracy at the reference's threshold
        assert abs(acc.mean() - oacc.mean()) * 100 <= 0.1 + 1e-9         # ... and with thresholds searched on each side


def test_accuracy_within_point_one_on_perturbed_features(device):
    """Features perturbed by the error level the fp16 path shows (cosine 0.9999 ~ 1.4 % relative
    noise is the allowed worst case; measured is ~10x smaller) keep 10-fold accuracy within 0.1 pt."""
    fl, fr, ds = synthetic.planted_features(6000, seed=21, same_cos=0.5, diff_cos=0.12)
    rng = np.random.default_rng(0)
    eps = 0.002
    fl2 = fl + eps * rng.standard_normal(fl.shape).astype(np.float32) / np.sqrt(512)
    fr2 = fr + eps * rng.standard_normal(fr.shape).astype(np.float32) / np.sqrt(512)
    for method in ("l2_distance", "cos_distance"):
        a, _ = Evaluation.evaluation_10_fold(fl, fr, ds, method)
        b, _ = Evaluation.evaluation_10_fold(fl2, fr2, ds, method)
        assert abs(a.mean() - b.mean()) * 100 <= 0.1


def test_device_10fold_equals_reference_golden(device, golden_dir):
    """Features resident on the GPU: thresholds and accuracies bit-identical to the reference's run."""
    z = np.load(os.path.join(golden_dir, "eval10fold.npz"))
    import ast
    for tag, pairs in (("p600", 600), ("p1200_hard", 1200), ("p6000", 6000)):
        seed = int(z[f"{tag}_seed"])                       # the seeds / arguments make_golden.py used
        kw = ast.literal_eval(str(z[f"{tag}_kw"]))
        fl, fr, ds = synthetic.planted_features(pairs, seed=seed, **kw)
        dl, dr = torch.from_numpy(fl).to(device), torch.from_numpy(fr).to(device)
        keep = dl.clone()
        for method in ("l2_distance", "cos_distance"):
            accs, thr = Evaluation.evaluation_10_fold(dl, dr, ds, method)
            assert np.array_equal(accs, z[f"{tag}_{method}_accs"]), (tag, method)
            assert np.array_equal(thr, z[f"{tag}_{method}_thr"]), (tag, method)
        assert torch.equal(dl, keep)          # the caller's tensors are left alone


@pytest.mark.parametrize("D", [128, 256, 512])
def test_device_fold_scores_bit_identical_to_numpy(device, D):
    """ev_fold_scores against the reference's array expressions (Evaluation.py:84-94), two folds in a row
    so that the carried-over centring is covered; ragged pair count."""
    from face_recognition_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(D)
    P = 1237
    fl = (rng.standard_normal((P, D)) * 0.2 + 0.05).astype(np.float32)
    fr = (fl * 0.5 + rng.standard_normal((P, D)) * 0.2).astype(np.float32)
    folds = rng.integers(0, 10, P)
    dl, dr = torch.from_numpy(fl).to(device), torch.from_numpy(fr).to(device)
    mu = torch.empty(D, device=device)
    sc = torch.empty(P, device=device)
    for method_id, method in ((0, "l2_distance"), (1, "cos_distance")):
        for i in (3, 4):
            val = folds != i
            m = np.mean(np.concatenate((fl[val], fr[val]), 0), 0)[None]
            fl, fr = fl - m, fr - m
            fl = fl / np.expand_dims(np.sqrt(np.sum(np.power(fl, 2), 1)), 1)
            fr = fr / np.expand_dims(np.sqrt(np.sum(np.power(fr, 2), 1)), 1)
            want = np.sum(np.power(fl - fr, 2), 1) if method_id == 0 else np.sum(np.multiply(fl, fr), 1)
            mask = torch.from_numpy(val.astype(np.uint8)).to(device)
            _lib.check(lib.ev_fold_scores(dl.data_ptr(), dr.data_ptr(), mask.data_ptr(), int(val.sum()), P, D, method_id,
                                          mu.data_ptr(), sc.data_ptr(), _lib.current_stream_ptr(device)), "ev_fold_scores")
            assert np.array_equal(sc.cpu().numpy(), want), (method, i)
            assert np.array_equal(dl.cpu().numpy(), fl) and np.array_equal(dr.cpu().numpy(), fr)
    with pytest.raises(RuntimeError, match="feature width"):
        _lib.check(lib.ev_fold_scores(dl.data_ptr(), dr.data_ptr(), mask.data_ptr(), 5, P, 100, 0, mu.data_ptr(),
                                      sc.data_ptr(), _lib.current_stream_ptr(device)), "ev_fold_scores")


def test_get_feature_on_device_feeds_device_10fold(hip_model, device):
    pairs = 40
    left, right, ds = synthetic.pair_crops(pairs, seed=9, noise=0.3)
    fl, fr = Evaluation.getFeature(hip_model, _loader(left, right, 20), device, flip=True, keep_on_device=True)
    assert fl.is_cuda and fr.is_cuda
    a_dev, t_dev = Evaluation.evaluation_10_fold(fl, fr, ds, "cos_distance")
    a_host, t_host = Evaluation.evaluation_10_fold(fl.cpu(), fr.cpu(), ds, "cos_distance")
    assert np.array_equal(a_dev, a_host) and np.array_equal(t_dev, t_host)


def test_device_features_of_other_widths_take_the_host_path(device):
    """embedding_size 1024 is a width the device scoring kernel does not take: evaluation_10_fold then runs the
    reference's host arithmetic on a copy instead of failing."""
    rng = np.random.default_rng(3)
    ds = synthetic.PairSet(200)
    fl = rng.standard_normal((200, 1024)).astype(np.float32)
    fr = (fl + 0.8 * rng.standard_normal((200, 1024))).astype(np.float32)
    a_dev, t_dev = Evaluation.evaluation_10_fold(torch.from_numpy(fl).to(device), torch.from_numpy(fr).to(device), ds, "cos_distance")
    a_host, t_host = Evaluation.evaluation_10_fold(fl, fr, ds, "cos_distance")
    assert np.array_equal(a_dev, a_host) and np.array_equal(t_dev, t_host)
