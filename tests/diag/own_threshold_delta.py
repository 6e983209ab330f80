"""Accuracy of the harness phases with thresholds searched on each side (the reference's protocol), GPU features against
the fp32 oracle's, for calibrated and for uncalibrated random weights.  python tests/diag/own_threshold_delta.py [pairs] [noise] [phases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from face_recognition_amd import synthetic                                    # noqa: E402
from face_recognition_amd import test_lfw_agedb_cfpfp as harness              # noqa: E402
from face_recognition_amd.Evaluation import getFeature                        # noqa: E402
from face_recognition_amd.face_model import MobileFaceNet                     # noqa: E402
from oracle import evaluation_ref, mobilefacenet_ref                          # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
phases = sys.argv[3].split(",") if len(sys.argv) > 3 else ["LFW", "CFP_FP", "AgeDB30"]
dev = torch.device("cuda", 0)
sd = synthetic.make_state_dict(1, 512)
left, right, _ = synthetic.pair_crops(60, seed=99, noise=0.45)
taps = {}
mobilefacenet_ref.forward(sd, torch.from_numpy(np.concatenate([left, right])), taps)
y = taps["pre_norm"].numpy().astype(np.float64)
g, b, mu, var = (np.asarray(sd["bn." + k], dtype=np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
z = (y - b) / g * np.sqrt(var + 1e-5) + mu
cal = synthetic.with_embedding_statistics(sd, z.mean(0), z.var(0))
for name, w in (("calibrated", cal), ("uncalibrated", sd)):
    m = MobileFaceNet(512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.to(dev).eval()
    sizes = {"LFW": pairs, "CFP_FP": pairs // 10 * 14, "AgeDB30": pairs}
    _, _, sets = harness.load_test_data(50, sizes=sizes, seed=5, noise=noise, device=dev)
    embed = lambda x: mobilefacenet_ref.forward(w, x)      # noqa: E731
    for phase, ds in sets.items():
        if phase not in phases:
            continue
        cpu_batches = ([t.cpu() for t in bb] for bb in harness.PairLoader(ds, 200))
        ol, orr = evaluation_ref.get_feature(embed, cpu_batches, flip=True)
        oacc, othr = evaluation_ref.evaluation_10_fold(ol.numpy(), orr.numpy(), ds.folds, ds.flags, "cos_distance")
        fl, fr = getFeature(m, harness.PairLoader(ds, 50), dev, flip=True)
        acc, thr = evaluation_ref.evaluation_10_fold(fl.numpy(), fr.numpy(), ds.folds, ds.flags, "cos_distance")
        at_ref, _ = evaluation_ref.evaluation_10_fold(fl.numpy(), fr.numpy(), ds.folds, ds.flags, "cos_distance", fixed_thresholds=othr)
        print("%-12s noise %.2f %-8s pairs %5d  oracle %.4f  gpu own thresholds %.4f (delta %+.4f pt, thr %+.4f)  at oracle thresholds %.4f (delta %+.4f pt)" % (
            name, noise, phase, len(ds.flags), oacc.mean() * 100, acc.mean() * 100, (acc.mean() - oacc.mean()) * 100, thr.mean() - othr.mean(),
            at_ref.mean() * 100, (at_ref.mean() - oacc.mean()) * 100), flush=True)
