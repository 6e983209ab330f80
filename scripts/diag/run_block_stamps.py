"""Run the stamped diagnostic library (build_ab/lib_stamps.so copied over the in-tree one by the
caller) and print the median phase timeline of one fused block kernel.
usage: run_block_stamps.py <n_ops> [batch]   -- n_ops = kernels of the forward to run; the last one is reported
"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd import _lib
from face_recognition_amd.face_model import MobileFaceNet
from face_recognition_amd import synthetic

n_ops = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to("cuda:0").eval()
x = torch.randn(B, 3, 112, 112, device="cuda:0")
for _ in range(3):
    net.debug_prefix(x, n_ops)
torch.cuda.synchronize()
lib = _lib.load()
NS = 64
lib.fra_dbg_read.restype = C.c_int
lib.fra_dbg_read.argtypes = [C.c_void_p, C.c_size_t]
nwg = int(sys.argv[3]) if len(sys.argv) > 3 else 256
buf = np.zeros((nwg, NS), dtype=np.uint64)
rc = lib.fra_dbg_read(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
st = buf.astype(np.int64)
t0 = st[:, 60:61]
rel = st - t0
used = [i for i in range(60) if (st[:, i] != 0).all()] + [61]
real = (st[:, 63] - st[:, 62]) * 10.0  # ns (100 MHz)
cyc = rel[:, 61]
print("workgroups", nwg, "median total cycles", np.median(cyc), "median wall ns", np.median(real),
      "=> clock GHz %.2f" % (np.median(cyc) / np.median(real)))
start_ns = (st[:, 62] - st[:, 62].min()) * 10.0
print("start times (us): min %.1f median %.1f max %.1f ; end max %.1f" % (start_ns.min() / 1e3, np.median(start_ns) / 1e3, start_ns.max() / 1e3, ((st[:, 63] - st[:, 62].min()) * 10.0).max() / 1e3))
prev = 0
for i in used:
    m = np.median(rel[:, i])
    print("  stamp %2d  median cycles %8.0f  (+%7.0f)  p10 %8.0f p90 %8.0f" % (i, m, m - prev, np.percentile(rel[:, i], 10), np.percentile(rel[:, i], 90)))
    prev = m
