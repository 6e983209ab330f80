"""Where the wide-wave chain kernel differs from the eight-wave one: python scripts/diag/wb_diff.py [B] [nblk]
(prints, per image row and per block count, how many values differ)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_recognition_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c, cexp, H = 128, 256, 14
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
pks = []
for _ in range(nmax):
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
per = pks[0].numel()
x = torch.randn((B, H, H, c), generator=g).half().to(dev)
s = torch.cuda.current_stream().cuda_stream
for nblk in range(1, nmax + 1):
    ref = x
    for i in range(nblk):                                   # one launch per block on the eight-wave kernel's bands
        nxt = torch.empty_like(x)
        _lib.check(lib.mfn_op_rblock(_lib.FRA_F16, c, cexp, H, ref.data_ptr(), pk.data_ptr() + i * per, 1, nxt.data_ptr(), B, s), "rblock")
        ref = nxt
    out = torch.full_like(x, float("nan"))
    _lib.check(lib.mfn_op_wblock(_lib.FRA_F16, c, cexp, H, x.data_ptr(), pk.data_ptr(), nblk, out.data_ptr(), B, s), "wblock")
    torch.cuda.synchronize()
    bad = (out != ref) | out.isnan()
    print(f"nblk={nblk}: {int(bad.sum())} of {bad.numel()} differ; per row {bad.sum(dim=(0, 2, 3)).tolist()}; "
          f"per column {bad.sum(dim=(0, 1, 3)).tolist()}; per image {bad.sum(dim=(1, 2, 3)).tolist()[:8]}")
    if bad.any():
        ch = bad.sum(dim=(0, 1, 2))
        print("   per channel (first 32):", ch[:32].tolist(), " max |diff|", float((out.float() - ref.float()).abs().nan_to_num(1e9).max()))
rc, tag = lib.fra_device_fault(0, None), None
print("device fault:", rc)
