"""Where the waves of the chained conv_4 kernel (kernels_rblock.hip built with -DRB_STAMPS) spend their cycles:
python scripts/diag/rb_stamps.py [batch]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd import _lib, synthetic
from face_recognition_amd.face_model import MobileFaceNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to("cuda:0").eval()
x = torch.randn(B, 3, 112, 112, device="cuda:0")
names = [r[0] for r in (net(x), net.op_table())[1]]
n = names.index("conv_4.model.5") + 1       # the last launch is the chain
for _ in range(3):
    net.debug_prefix(x, n)
torch.cuda.synchronize()
lib = _lib.load()
lib.fra_dbg_read.restype = C.c_int
lib.fra_dbg_read.argtypes = [C.c_void_p, C.c_size_t]
nwg = min(B, 256)
buf = np.zeros((nwg, 8, 8), dtype=np.uint64)
assert lib.fra_dbg_read(buf.ctypes.data, buf.nbytes) == 0
c = buf.astype(np.float64)
print("median cycles per wave over %d workgroups (kernel total %.0f)" % (nwg, np.median(c[:, :, 7])))
# chained launch (flags): what each counter holds; the per-block launches with barriers count
# expand | barrier after expand | depthwise + project | block epilogue + package wait | barrier after the slice | package request
parts = ("expand + flag", "wait: package, neighbours done with slice s - 2", "depthwise + project", "wait: neighbours' rows", "block epilogue + flag", "-")
for w in range(8):
    m = np.median(c[:, w, :], axis=0)
    print("  wave %d: %s" % (w, ", ".join("%s %.0f" % (p, v) for p, v in zip(parts, m[:6]))), " total %.0f" % m[7])
