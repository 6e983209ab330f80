"""Where the waves of the wide-wave chain kernel (kernels_wblock.hip built with -DWB_STAMPS) spend their cycles:
FRA_LIB_PATH=build_ab/wb_stamps.so python scripts/diag/wb_stamps.py [batch] [blocks]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c, cexp, H = 128, 256, 14
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
pk = torch.randint(0, 255, (nblk * lib.mfn_rblock_package_bytes(c, cexp),), dtype=torch.uint8, generator=g)
pk16 = pk.view(torch.float16)
pk16[pk16.isnan() | pk16.isinf() | (pk16.abs() > 4)] = 0.01      # timing only: any finite weights
pk = pk.to(dev)
x = torch.randn((B, H, H, c), generator=g).half().to(dev)
out = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    _lib.check(lib.mfn_op_wblock(_lib.FRA_F16, c, cexp, H, x.data_ptr(), pk.data_ptr(), nblk, out.data_ptr(), B, s), "wblock")
torch.cuda.synchronize()
lib.fra_dbg_read_wb.restype = C.c_int
lib.fra_dbg_read_wb.argtypes = [C.c_void_p, C.c_size_t]
nwg = min(B, 256)
buf = np.zeros((nwg, 4, 8), dtype=np.uint64)
assert lib.fra_dbg_read_wb(buf.ctypes.data, buf.nbytes) == 0
cyc = buf.astype(np.float64)
per = (B + nwg - 1) // nwg * nblk * (cexp // 64)
print("median cycles per wave over %d workgroups: kernel total %.0f, %d slices per workgroup -> per slice:" % (nwg, np.median(cyc[:, :, 7]), per))
parts = ("wait: package + neighbours' reads", "expand + flag", "wait: neighbours' rows", "package request", "depthwise + project", "epilogue + drain", "-")
for w in range(4):
    m = np.median(cyc[:, w, :], axis=0)
    print("  wave %d: %s" % (w, ", ".join("%s %.0f" % (p, v / per) for p, v in zip(parts, m[:6]))), " total/slice %.0f" % (m[7] / per))
