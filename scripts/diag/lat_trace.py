"""Batch-1 forward + match replayed as a hipGraph, for a rocprofv3 --kernel-trace run: prints nothing, the trace is the result."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd.face_model import MobileFaceNet
from face_recognition_amd.graphed import GraphedRecognizer
from face_recognition_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to("cuda:0").eval()
bank = torch.from_numpy(synthetic.unit_rows(1000, 512, seed=2)).to("cuda:0")
rec = GraphedRecognizer(net, bank, tta=False, threshold=60)
x = torch.randn(B, 3, 112, 112, device="cuda:0")
for _ in range(20):
    rec(x)
torch.cuda.synchronize()
for _ in range(5):
    net(x)
torch.cuda.synchronize()
