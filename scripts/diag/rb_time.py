"""Time the register-resident residual-block launch alone: python scripts/diag/rb_time.py [B] [nblk] [reps]
(random weights and activations; microseconds per launch by HIP events on the launch stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
c, cexp, H = 128, 256, 14
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
pks = []
for _ in range(nblk):
    w1 = torch.randn((cexp, c), generator=g) / np.sqrt(c) * 1.3
    w2 = torch.randn((cexp, 9), generator=g) * 0.4
    w3 = torch.randn((c, cexp), generator=g) / np.sqrt(cexp) * 0.6
    b1, b2, b3 = (torch.randn(n, generator=g) * 0.2 for n in (cexp, cexp, c))
    s1, s2 = (torch.rand(cexp, generator=g) * 0.3 + 0.1 for _ in range(2))
    pk = torch.zeros(lib.mfn_rblock_package_bytes(c, cexp), dtype=torch.uint8)
    _lib.check(lib.mfn_pack_rblock(_lib.FRA_F16, c, cexp, *[t.contiguous().data_ptr() for t in (w1, b1, s1, w2, b2, s2, w3, b3)],
                                   pk.data_ptr()), "pack")
    pks.append(pk)
pk = torch.cat(pks).to(dev)
x = torch.randn((B, H, H, c), generator=g).half().to(dev)
out = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream


def run():
    _lib.check(lib.mfn_op_rblock(_lib.FRA_F16, c, cexp, H, x.data_ptr(), pk.data_ptr(), nblk, out.data_ptr(), B, s), "rblock")


for _ in range(30):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"rblock B={B} nblk={nblk}: {us:.1f} us per launch, {us / nblk:.2f} us per block, "
      f"{us * 1e-6 * 2.1e9 / nblk / max(1, (B + 255) // 256):.0f} cycles per image-block at 2.1 GHz")
