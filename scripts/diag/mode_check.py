"""Small-batch mode 1 against mode 0 at the given batch sizes: min cosine per batch, and whether mode-1 rows repeat bit for bit
between batch sizes: python scripts/diag/mode_check.py 1 8 36 37 64 73"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import face_model, synthetic
from face_recognition_amd.face_model import MobileFaceNet
dev = torch.device("cuda", 0)
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to(dev).eval()
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 36, 37, 64, 73]
x = torch.from_numpy(synthetic.make_crops(max(sizes), 77)).to(dev)
for B in sizes:
    face_model.set_small_batch_mode(1)
    a = net(x[:B].contiguous()).clone()
    a2 = net(x[:B].contiguous()).clone()
    face_model.set_small_batch_mode(0)
    b = net(x[:B].contiguous()).clone()
    cos = (a * b).sum(1)
    print(f"B={B:4d}: mode 1 vs mode 0 min cosine {float(cos.min()):.7f}  equal bits {bool(torch.equal(a, b))}  mode 1 repeatable {bool(torch.equal(a, a2))}", flush=True)
face_model.set_small_batch_mode(1)
