"""Embeddings of several batch sizes to a file (run once with the default kernel choice and once with FRA_PBLOCK_MIN=0,
then compare): python scripts/diag/pb_equiv.py <out.npz>   |   python scripts/diag/pb_equiv.py --compare a.npz b.npz"""
import sys, numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k]) and np.isfinite(a[k]).all()
        ok &= same
        print(k, "identical" if same else "DIFFERENT", a[k].shape)
    sys.exit(0 if ok else 1)
import torch
sys.path.insert(0, ".")
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(1, 512)
net = MobileFaceNet(512); net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net = net.to(dev).eval()
base = torch.from_numpy(synthetic.make_crops(128, 31)).to(dev)
out = {}
for B in (250, 255, 256, 257, 477, 500, 512, 704, 931, 1000, 1023, 1024, 2040, 2048, 3000):
    x = base.repeat((B + 127) // 128, 1, 1, 1)[:B].contiguous()
    x += torch.linspace(-0.2, 0.2, B, device=dev).view(B, 1, 1, 1)
    net(x)
    v = net.op_variants(B)
    y = net(x).cpu().numpy()
    out["B%d" % B] = y
    print(B, "pipeline" if 2 in v else "bands", flush=True)
np.savez(sys.argv[1], **out)
