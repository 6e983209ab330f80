"""What the host side of a batch-1 graph replay costs: staging copy or not, how the host waits.
python scripts/diag/lat_wait.py [batch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
from face_recognition_amd.graphed import GraphedRecognizer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to(dev).eval()
bank = torch.from_numpy(synthetic.unit_rows(1000, 512, seed=2)).to(dev)
rec = GraphedRecognizer(net, bank, tta=False, threshold=60)
x = torch.rand((B, 3, 112, 112), device=dev) * 2 - 1
rec(x)
cap = rec._graphs[B]
ev = torch.cuda.Event()
st = torch.cuda.current_stream(dev)


def p50(fn, reps=1000, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e6
    return "p50 %.1f  p99 %.1f" % (ts[len(ts) // 2], ts[int(len(ts) * 0.99)])


def spin():
    ev.record(st)
    while not ev.query():
        pass


print("copy + replay + device synchronize   ", p50(lambda: (rec(x), torch.cuda.synchronize())))
print("copy + replay + stream synchronize   ", p50(lambda: (rec(x), st.synchronize())))
print("replay + device synchronize          ", p50(lambda: (cap.graph.replay(), torch.cuda.synchronize())))
print("replay + stream synchronize          ", p50(lambda: (cap.graph.replay(), st.synchronize())))
print("replay + event record + query loop   ", p50(lambda: (cap.graph.replay(), spin())))
print("copy + replay + event query loop     ", p50(lambda: (rec(x), spin())))
print("eager forward + device synchronize   ", p50(lambda: (net(x), torch.cuda.synchronize()), 300, 30))
print("eager forward + event query loop     ", p50(lambda: (net(x), spin()), 300, 30))
print("replay only (host time of the launch)", p50(lambda: cap.graph.replay(), 200, 20))
torch.cuda.synchronize()
