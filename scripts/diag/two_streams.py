"""Throughput of the forward with the batch split over two streams (two model handles), against one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd.face_model import MobileFaceNet
from face_recognition_amd import synthetic
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()}
nets = []
for _ in range(4):
    n = MobileFaceNet(512); n.load_state_dict(sd); nets.append(n.to(dev).eval())
B = int(os.environ.get("FRA_B", "1024"))
x = torch.randn(B, 3, 112, 112, device=dev)
def run(parts, steps=150):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = list(torch.chunk(x, parts))
    def step():
        for s, n, xx in zip(streams, nets, xs):
            with torch.cuda.stream(s):
                n(xx)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("parts %d: %.4f ms/step, %.0f faces/s" % (parts, dt * 1e3, B / dt), flush=True)
for p in (1, 2, 1, 2, 1, 2, 4, 1):
    run(p)
