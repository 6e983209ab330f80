"""Repeat small-batch forwards and count results that differ from the first one (a race in the flag / barrier protocols of the
band launches would show as a rare mismatch): python scripts/diag/soak_small_batch.py [iterations]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import face_model, synthetic
from face_recognition_amd.face_model import MobileFaceNet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to(dev).eval()
bad_total = 0
for mode in (1, 0):
    face_model.set_small_batch_mode(mode)
    for B in (1, 2, 3, 5, 8, 16, 35, 36, 37, 64, 256):
        x = torch.from_numpy(synthetic.make_crops(B, 40 + B)).to(dev)
        first = net(x).clone()
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        n = iters if B <= 64 else iters // 4
        for i in range(n):
            y = net(x)
            bad += (y != first).any()
        torch.cuda.synchronize()
        print(f"mode {mode} B={B:4d}: {int(bad)} of {n} forwards differ from the first", flush=True)
        bad_total += int(bad)
face_model.set_small_batch_mode(1)
print("TOTAL mismatches", bad_total)
sys.exit(1 if bad_total else 0)
