#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run once in the build container, where the reference checkout is mounted at
/root/reference (it does not exist on the GPU box):

    python tests/golden/make_golden.py

What it does
  1. imports the reference's ``face_model.MobileFaceNet`` (torch-only module), loads the
     seeded synthetic state_dict from ``face_recognition_amd.synthetic`` and records its
     outputs on seeded crops                                  -> embeddings_b4.npz
  2. imports the reference's ``Evaluation`` module -- registering empty stand-in modules for
     ``cv2`` / ``torchvision`` / ``data_set.dataloader`` first, because Evaluation.py imports
     them at the top but its four functions never touch them -- and records
     ``evaluation_10_fold`` / ``getThreshold`` / ``getAccuracy`` results on seeded features
                                                              -> eval10fold.npz
  3. loads the ``facebank.pth`` + ``names.npy`` pair the reference ships (weights_only=True,
     allow_pickle=False) and records the reference's inline match expressions on it
                                                              -> facebank_match.npz
     (``facebank.py`` itself needs cv2 and MTCNN at import time and cannot be imported here;
     the match lines of cam_demo.py:97-105 are evaluated through oracle.matching_ref, which
     is those same expressions.)
  4. cross-checks the CPU restatements under ``oracle/`` against every reference result.
  5. (``--only calibrated`` runs this step alone) measures, with the REFERENCE model, the mean and variance of the
     ``linear`` layer's output over 120 verification-style crops, sets the final BatchNorm's running statistics to them
     -- as a trained network has them: embeddings then spread over the sphere instead of clustering around one direction,
     which is the regime in which a per-embedding cosine is an honest figure -- and records the reference's embeddings of
     16 further crops with those weights                        -> embeddings_calibrated.npz

Inputs and weights are regenerated from their seeds at test time, so only seeds and expected
outputs are stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Mobilefacenet"
sys.path.insert(0, ROOT)

from face_recognition_amd import layers, synthetic  # noqa: E402
from oracle import evaluation_ref, matching_ref, mobilefacenet_ref  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    import face_model as ref_model  # torch + math only
    for name in ("cv2", "torchvision", "torchvision.transforms", "data_set", "data_set.dataloader"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    dl = sys.modules["data_set.dataloader"]
    dl.LFW = dl.CFP_FP = dl.AgeDB30 = object
    import Evaluation as ref_eval
    return ref_model, ref_eval


def calibrated(ref_model):
    sd_np = synthetic.make_state_dict(1, 512)
    net = ref_model.MobileFaceNet(512).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    taps = []
    hook = net.linear.register_forward_hook(lambda mod, inp, out: taps.append(out.detach().double()))
    left, right, _ = synthetic.pair_crops(60, seed=99, noise=0.45)
    with torch.no_grad():
        net(torch.from_numpy(np.concatenate([left, right])))
    hook.remove()
    z = torch.cat(taps)
    mean, var = z.mean(0).numpy(), z.var(0, unbiased=False).numpy()
    sd_cal = synthetic.with_embedding_statistics(sd_np, mean, var)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_cal.items()})
    l2, r2, _ = synthetic.pair_crops(20, seed=123, noise=0.3)
    x = torch.from_numpy(np.concatenate([l2[:8], r2[:8]]))
    with torch.no_grad():
        y = net(x)
        y_tta = ref_model.l2_norm(y + net(torch.flip(x, dims=[3])))
    o = mobilefacenet_ref.forward(sd_cal, x)
    cosm = (y @ y.t()).numpy()
    off = cosm[~np.eye(len(cosm), dtype=bool)]
    print("calibrated: oracle vs reference max|d| %.3g; pairwise cosine of the 16 embeddings: mean %.3f, max %.3f" %
          ((o - y).abs().max(), off.mean(), off.max()))
    assert (o - y).abs().max() < 2e-6
    np.savez_compressed(os.path.join(HERE, "embeddings_calibrated.npz"), weight_seed=1, stat_pairs=60, stat_seed=99,
                        stat_noise=0.45, bn_running_mean=mean.astype(np.float32), bn_running_var=var.astype(np.float32),
                        crop_pairs=20, crops_used_per_side=8, crop_seed=123, crop_noise=0.3, emb=y.numpy(), emb_tta=y_tta.numpy())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_eval = import_reference()
    if sys.argv[1:3] == ["--only", "calibrated"]:
        calibrated(ref_model)
        return

    # ---- 1. embeddings ------------------------------------------------------------------
    W_SEED, X_SEED, B = 1, 0, 4
    sd_np = synthetic.make_state_dict(W_SEED, 512)
    net = ref_model.MobileFaceNet(512).eval()
    ref_keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    my_keys = [(k, tuple(s)) for k, s, _ in layers.state_dict_keys(512)]
    assert ref_keys == my_keys, "state_dict keys/shapes/order differ from the reference"
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    x = torch.from_numpy(synthetic.make_crops(B, X_SEED))
    xs = torch.from_numpy(synthetic.make_smooth_crops(B, X_SEED))
    with torch.no_grad():
        y = net(x)
        y_flip = net(torch.flip(x, dims=[3]))
        y_tta = ref_model.l2_norm(y + y_flip)
        ys = net(xs)
    o = mobilefacenet_ref.forward(sd_np, x)
    o_tta = mobilefacenet_ref.embed_tta(sd_np, x)
    print("oracle vs reference  max|d| emb %.3g  tta %.3g" % ((o - y).abs().max(), (o_tta - y_tta).abs().max()))
    assert (o - y).abs().max() < 2e-6 and (o_tta - y_tta).abs().max() < 2e-6
    # second embedding size, exercises the embedding_size argument
    sd128 = synthetic.make_state_dict(7, 128)
    net128 = ref_model.MobileFaceNet(128).eval()
    net128.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd128.items()})
    with torch.no_grad():
        y128 = net128(x[:2])
    assert (mobilefacenet_ref.forward(sd128, x[:2]) - y128).abs().max() < 2e-6
    np.savez_compressed(os.path.join(HERE, "embeddings_b4.npz"),
                        weight_seed=W_SEED, input_seed=X_SEED, batch=B,
                        emb=y.numpy(), emb_flip=y_flip.numpy(), emb_tta=y_tta.numpy(),
                        emb_smooth=ys.numpy(), emb128_seed7_b2=y128.numpy())

    # ---- 2. evaluation --------------------------------------------------------------------
    out = {}
    for tag, pairs, seed, kw in (("p600", 600, 5, {}), ("p1200_hard", 1200, 11, dict(same_cos=0.4, diff_cos=0.2)),
                                 ("p6000", 6000, 21, dict(same_cos=0.5, diff_cos=0.12))):
        fl, fr, ds = synthetic.planted_features(pairs, seed=seed, **kw)
        for method in ("l2_distance", "cos_distance"):
            accs, thr = ref_eval.evaluation_10_fold(torch.from_numpy(fl), torch.from_numpy(fr), ds, method)
            o_accs, o_thr = evaluation_ref.evaluation_10_fold(fl, fr, ds.folds, ds.flags, method)
            assert np.array_equal(accs, o_accs) and np.array_equal(thr, o_thr), (tag, method)
            out[f"{tag}_{method}_accs"] = accs
            out[f"{tag}_{method}_thr"] = thr
            out[f"{tag}_seed"] = seed
            out[f"{tag}_kw"] = np.array(repr(kw))
            print(tag, method, "acc %.4f thr %.4f" % (accs.mean() * 100, thr.mean()))
    # getThreshold / getAccuracy edge cases: ties at the threshold grid, plateaus of equal accuracy
    rng = np.random.default_rng(3)
    grid = np.arange(-50, 51) * 3.0 / 50
    scores = np.concatenate([grid[rng.integers(0, 101, 40)], rng.uniform(-1, 1, 40)]).astype(np.float32)
    flags = np.where(rng.uniform(size=80) < 0.5, 1, -1)
    for method in ("l2_distance", "cos_distance"):
        t = ref_eval.getThreshold(scores, flags, 50, method)
        a = np.array([ref_eval.getAccuracy(scores, flags, g, method) for g in grid])
        assert t == evaluation_ref.get_threshold(scores, flags, 50, method)
        out[f"edge_{method}_thr"] = t
        out[f"edge_{method}_acc_curve"] = a
    out["edge_scores"] = scores
    out["edge_flags"] = flags
    np.savez_compressed(os.path.join(HERE, "eval10fold.npz"), **out)

    # ---- 3. the facebank the reference ships -----------------------------------------------
    fb = torch.load(os.path.join(REF, "facebank", "facebank.pth"), weights_only=True)
    names = np.load(os.path.join(REF, "facebank", "names.npy"), allow_pickle=False)
    assert fb.shape == (5, 512) and fb.dtype == torch.float32
    rng = np.random.default_rng(9)
    noise = torch.from_numpy(rng.standard_normal((5, 512)).astype(np.float32))
    q_near = ref_model.l2_norm(fb + 0.02 * noise)          # close to each identity -> accepted
    q_far = ref_model.l2_norm(fb + 0.111 * noise)          # d2 around the 1.2 / 1.325 rejection thresholds
    q_rand = torch.from_numpy(synthetic.unit_rows(3, 512, seed=10))
    q = torch.cat([fb, q_near, q_far, q_rand])
    res = {}
    for th in (60.0, 50.0):
        mn, idx, sc = matching_ref.match_broadcast(q, fb, th)
        res[f"min_th{int(th)}"] = mn.numpy()
        res[f"idx_th{int(th)}"] = idx.numpy()
        res[f"score_th{int(th)}"] = sc.numpy()
    d2 = ((fb[:, None, :] - fb[None, :, :]) ** 2).sum(-1)
    print("shipped facebank pairwise d2 off-diagonal range: %.4f .. %.4f" %
          (d2[~torch.eye(5, dtype=bool)].min(), d2.max()))
    print("match idx th60:", res["idx_th60"])
    np.savez_compressed(os.path.join(HERE, "facebank_match.npz"), facebank=fb.numpy(), names=names,
                        queries=q.numpy(), pairwise_d2=d2.numpy(), **res)
    print("wrote fixtures to", HERE)
    calibrated(ref_model)


if __name__ == "__main__":
    main()
