"""Forward time and per-kernel times over batch sizes: python scripts/diag/batch_sweep.py 1 8 37 64 128 200 256 512 1024"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
dev = torch.device("cuda", 0)
sd = synthetic.make_state_dict(1, 512)
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net = net.to(dev).eval()
detail = os.environ.get("SWEEP_DETAIL")
for B in [int(a) for a in sys.argv[1:]]:
    x = torch.rand((B, 3, 112, 112), device=dev) * 2 - 1
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        net(x)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    line = f"B={B:5d}: {us:8.1f} us  {B / us * 1e6:10.0f} faces/s"
    if detail:
        acc = None
        for _ in range(5):
            for _ in range(3):
                net(x)
            _, ms, table = net.forward_timed(x)
            acc = ms if acc is None else acc + ms
        ms = acc / 5 * 1e3
        groups = {}
        for (name, kind, *_), t in zip(table, ms):
            key = name.split(".model")[0] if kind == "block" else kind
            groups[key] = groups.get(key, 0.0) + t
        line += "  | " + "  ".join(f"{k} {v:.0f}" for k, v in groups.items())
    print(line, flush=True)
