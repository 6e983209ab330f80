"""Per-embedding cosine GPU vs fp32 oracle with weights whose final BatchNorm statistics match the activations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
from oracle import mobilefacenet_ref
sd = synthetic.make_state_dict(1, 512)
left, right, _ = synthetic.pair_crops(60, seed=99, noise=0.45)
taps = {}
mobilefacenet_ref.forward(sd, torch.from_numpy(np.concatenate([left, right])), taps)
y = taps["pre_norm"].numpy().astype(np.float64)
g, b, mu, var = (np.asarray(sd["bn." + k], dtype=np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
z = (y - b) / g * np.sqrt(var + 1e-5) + mu
cal = synthetic.with_embedding_statistics(sd, z.mean(0), z.var(0))
x = synthetic.pair_crops(100, seed=7, noise=0.45)[0]
dev = torch.device("cuda", 0)
for name, s_ in (("raw", sd), ("calibrated", cal)):
    for dt in ("f16", "bf16"):
        m = MobileFaceNet(512, act_dtype=dt)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in s_.items()})
        m = m.to(dev).eval()
        got = m(torch.from_numpy(x).to(dev)).cpu().numpy().astype(np.float64)
        taps = {}
        want = mobilefacenet_ref.forward(s_, torch.from_numpy(x), taps).numpy().astype(np.float64)
        cos = (got * want).sum(1)
        # error of the un-normalised output relative to its norm
        print(f"{name} {dt}: cosine min {cos.min():.6f} mean {cos.mean():.6f}; |mean embedding| {np.linalg.norm(want.mean(0)):.3f}", flush=True)
