"""Pair-score differences GPU vs fp32 oracle (cosine of TTA features, before the protocol's centring)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import synthetic, Evaluation
from face_recognition_amd import test_lfw_agedb_cfpfp as harness
from face_recognition_amd.face_model import MobileFaceNet
from oracle import mobilefacenet_ref, evaluation_ref
sd = synthetic.make_state_dict(1, 512)
left, right, _ = synthetic.pair_crops(60, seed=99, noise=0.45)
taps = {}
mobilefacenet_ref.forward(sd, torch.from_numpy(np.concatenate([left, right])), taps)
y = taps["pre_norm"].numpy().astype(np.float64)
g, b, mu, var = (np.asarray(sd["bn." + k], dtype=np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
z = (y - b) / g * np.sqrt(var + 1e-5) + mu
cal = synthetic.with_embedding_statistics(sd, z.mean(0), z.var(0))
dev = torch.device("cuda", 0)
m = MobileFaceNet(512)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in cal.items()})
m = m.to(dev).eval()
ds = harness.SyntheticPairs(800, seed=22, noise=0.45, device=dev)
fl, fr = Evaluation.getFeature(m, harness.PairLoader(ds, 50), dev, flip=True)
embed = lambda x: mobilefacenet_ref.forward(cal, x)
ol, orr = evaluation_ref.get_feature(embed, ([t.cpu() for t in bb] for bb in harness.PairLoader(ds, 200)), flip=True)
fl, fr, ol, orr = (t.numpy().astype(np.float64) for t in (fl, fr, ol, orr))
sg, so = (fl * fr).sum(1), (ol * orr).sum(1)
flags = np.asarray(ds.flags)
cosL = (fl * ol).sum(1)
print("feature cosine GPU vs oracle: min %.6f mean %.6f" % (cosL.min(), cosL.mean()))
for name, sel in (("same", flags == 1), ("different", flags == -1)):
    d = sg[sel] - so[sel]
    print(f"{name}: score mean {so[sel].mean():.4f} std {so[sel].std():.4f}; GPU - oracle: mean {d.mean():+.5f} std {d.std():.5f} max |d| {np.abs(d).max():.5f}")
