"""Where the waves of conv_4's small-batch launch (two-row bands, two blocks per launch; kernels_rblock.hip built with
-DRB_STAMPS) spend their cycles: python scripts/diag/rb_band_stamps.py [batch <= 36]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd import _lib, synthetic
from face_recognition_amd.face_model import MobileFaceNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = MobileFaceNet(512)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(1, 512).items()})
net = net.to("cuda:0").eval()
x = torch.randn(B, 3, 112, 112, device="cuda:0")
names = [r[0] for r in (net(x), net.op_table())[1]]
n = names.index("conv_4.model.5") + 1       # the last launch covers blocks 4 and 5
for _ in range(5):
    net.debug_prefix(x, n)
torch.cuda.synchronize()
lib = _lib.load()
lib.fra_dbg_read.restype = C.c_int
lib.fra_dbg_read.argtypes = [C.c_void_p, C.c_size_t]
nwg = 7 * B
buf = np.zeros((nwg, 8, 8), dtype=np.uint64)
assert lib.fra_dbg_read(buf.ctypes.data, buf.nbytes) == 0
c = buf.astype(np.float64)
print("median cycles per wave over %d workgroups, 8 slices (kernel total %.0f)" % (nwg, np.median(c[:, :, 7])))
parts = ("expand", "barrier after expand", "depthwise + project", "block epilogue + package wait", "barrier after the slice", "package request")
for w in range(8):
    m = np.median(c[:, w, :], axis=0)
    print("  wave %d: %s" % (w, ", ".join("%s %.0f" % (p, v) for p, v in zip(parts, m[:6]))), " set-up %.0f  loops %.0f" % (m[6], m[7]))
