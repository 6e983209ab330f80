"""Per-kernel milliseconds of the forward at a small batch (HIP events inside the library)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd.face_model import MobileFaceNet
from face_recognition_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to("cuda:0").eval()
x = torch.randn(B, 3, 112, 112, device="cuda:0")
acc = None
for i in range(30):
    _, ms, table = net.forward_timed(x)
    if i >= 10:
        acc = ms if acc is None else acc + ms
acc /= 20
for (name, kind, *_), m in zip(table, acc):
    print("%-28s %-8s %7.1f us" % (name, kind, m * 1e3))
print("sum %.1f us" % (acc.sum() * 1e3))
