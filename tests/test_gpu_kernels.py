"""Each HIP kernel, called through the C ABI, against the same op in plain PyTorch fp32 on the
same (16-bit rounded) operands.  Tolerances: the kernels accumulate in fp32 and round once to the
16-bit output, so the error bound is one output rounding (2^-11 relative for f16, 2^-8 for bf16)
plus fp32 summation-order noise."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from face_recognition_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f16": (torch.float16, _lib.FRA_F16, 2.0 ** -10), "bf16": (torch.bfloat16, _lib.FRA_BF16, 2.0 ** -7)}


def stream(dev):
    return _lib.current_stream_ptr(dev)


def close(out, ref, rel, what):
    out = out.float().cpu()
    ref = ref.float().cpu()
    err = (out - ref).abs()
    tol = rel * ref.abs() + rel * ref.abs().max() * 0.05 + 1e-6
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.4g} (ref max {ref.abs().max():.4g})"


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B", [1, 3])
def test_stem(device, dt, B):
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    x = torch.rand((B, 3, 112, 112), generator=g) * 2 - 1
    w = torch.randn((64, 3, 3, 3), generator=g) * 0.2
    bias = torch.randn(64, generator=g) * 0.1
    slope = torch.rand(64, generator=g) * 0.3 + 0.1
    ref = F.prelu(F.conv2d(x, w, bias, 2, 1), slope).permute(0, 2, 3, 1).contiguous()
    wk = w.reshape(64, 27).t().contiguous()       # [tap][oc]
    xd, wd, bd, sd = (t.to(device) for t in (x, wk, bias, slope))
    out = torch.empty((B, 56, 56, 64), dtype=tdt, device=device)
    _lib.check(lib.mfn_op_stem(code, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), sd.data_ptr(), out.data_ptr(), B,
                               stream(device)), "stem")
    torch.cuda.synchronize()
    close(out, ref, rel, "stem")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,C,stride,prelu", [(2, 56, 64, 1, True), (1, 56, 128, 2, True), (3, 28, 128, 1, True),
                                                (2, 28, 256, 2, True), (2, 14, 256, 1, True), (1, 14, 512, 2, True),
                                                (5, 7, 256, 1, True), (2, 14, 64, 1, False)])
def test_dw3x3(device, dt, B, H, C, stride, prelu):
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    x = (torch.randn((B, C, H, H), generator=g)).to(tdt).float()
    w = (torch.randn((C, 1, 3, 3), generator=g) * 0.3).to(tdt).float()
    bias = torch.randn(C, generator=g) * 0.1
    slope = torch.rand(C, generator=g) * 0.3 + 0.1
    y = F.conv2d(x, w, bias, stride, 1, 1, C)
    ref = (F.prelu(y, slope) if prelu else y).permute(0, 2, 3, 1).contiguous()
    Ho = ref.shape[1]
    xin = x.permute(0, 2, 3, 1).contiguous().to(tdt).to(device)
    wk = w.reshape(C, 9).t().contiguous().to(tdt).to(device)      # [tap][c]
    bd, sd = bias.to(device), slope.to(device)
    out = torch.empty((B, Ho, Ho, C), dtype=tdt, device=device)
    _lib.check(lib.mfn_op_dw3x3(code, xin.data_ptr(), wk.data_ptr(), bd.data_ptr(), sd.data_ptr() if prelu else None,
                                out.data_ptr(), B, H, H, C, stride, stream(device)), "dw3x3")
    torch.cuda.synchronize()
    close(out, ref, rel, f"dw3x3 H={H} C={C} s={stride}")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("M,K,N,prelu,resid", [(3136, 64, 128, True, False), (784, 128, 64, False, True),
                                               (784 * 2 + 5, 64, 256, True, False), (196, 256, 128, False, True),
                                               (196 * 3, 128, 512, True, False), (49, 512, 128, False, False),
                                               (49 * 5, 128, 256, True, False), (49, 256, 128, False, True),
                                               (1, 64, 64, False, False), (130, 128, 192, True, True)])
def test_pw(device, dt, M, K, N, prelu, resid):
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    a = torch.randn((M, K), generator=g).to(tdt)
    w = (torch.randn((N, K), generator=g) / np.sqrt(K)).to(tdt)
    bias = torch.randn(N, generator=g) * 0.1
    slope = torch.rand(N, generator=g) * 0.3 + 0.1
    r = torch.randn((M, N), generator=g).to(tdt)
    ref = a.float() @ w.float().t() + bias
    if prelu:
        ref = torch.where(ref > 0, ref, ref * slope)
    if resid:
        ref = ref + r.float()
    ad, wd, bd, sd, rd = (t.to(device) for t in (a, w, bias, slope, r))
    out = torch.full((M, N), float("nan"), dtype=tdt, device=device)
    _lib.check(lib.mfn_op_pw(code, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), sd.data_ptr() if prelu else None,
                             rd.data_ptr() if resid else None, out.data_ptr(), M, K, N, stream(device)), "pw")
    torch.cuda.synchronize()
    close(out, ref, rel, f"pw M={M} K={K} N={N}")


def _perm(c):
    """Permuted position of natural expanded channel c = 16g + 8a + 4h + i: 16g + 8h + 4a + i."""
    return (c & ~15) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 3)


def _w3_image(w3pos, cout):
    """Project weights [cout, cexp] (columns in position order) -> the slice-major order mfn_op_block reads
    (include/face_recognition_amd.h)."""
    cexp = w3pos.shape[1]
    co = torch.arange(cout).view(-1, 1).expand(cout, cexp)
    pos = torch.arange(cexp).view(1, -1).expand(cout, cexp)
    c, P, hi, g, i = pos // 64, (pos // 32) % 2, (pos // 16) % 2, (pos // 4) % 4, pos % 4
    idx = ((c * 2 * cout + 32 * (co // 16) + 16 * (g % 2) + co % 16) * 4 + 2 * P + g // 2) * 8 + 4 * hi + i
    out = torch.empty(cout * cexp, dtype=w3pos.dtype)
    out[idx.reshape(-1)] = w3pos.reshape(-1)
    assert idx.unique().numel() == cout * cexp
    return out


def _rblock_packages(lib, code, c, cexp, w1, b1, s1, w2, b2, s2, w3, b3):
    """Host packer of the library: folded weights in natural order -> the per-slice packages mfn_op_rblock streams."""
    n = lib.mfn_rblock_package_bytes(c, cexp)
    assert n > 0
    pk = torch.zeros(n, dtype=torch.uint8)
    arrs = [t.contiguous().float() for t in (w1, b1, s1, w2.reshape(cexp, 9), b2, s2, w3, b3)]
    _lib.check(lib.mfn_pack_rblock(code, c, cexp, *[a.data_ptr() for a in arrs], pk.data_ptr()), "pack_rblock")
    return pk


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("H,B", [(14, 1), (14, 2), (14, 36), (14, 37), (14, 129), (14, 300), (7, 2), (7, 37), (7, 131), (7, 300), (7, 515), (7, 1030)])
def test_rblock_chain(device, dt, H, B):
    """Three residual blocks (128 -> 256 -> 128 on 14 x 14 or 7 x 7) back to back: one launch with the activation kept
    in registers == three launches bit for bit, for every configuration the batch size selects (14 x 14: 2-row bands,
    7-row bands, whole images; 7 x 7: one-row bands, workgroups of one image or of two or four stacked images, batches that are no
    multiple of those), a small batch's images come out the same bits inside a large one, and all match the torch
    convs on 16-bit-rounded operands."""
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    c, cexp, nblk = 128, 256, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, c, H, H), generator=g).to(tdt).float()
    pks, y = [], x
    for _ in range(nblk):
        w1 = (torch.randn((cexp, c), generator=g) / np.sqrt(c) * 1.3).to(tdt).float()
        b1, s1 = torch.randn(cexp, generator=g) * 0.2, torch.rand(cexp, generator=g) * 0.3 + 0.1
        w2 = (torch.randn((cexp, 1, 3, 3), generator=g) * 0.4).to(tdt).float()
        b2, s2 = torch.randn(cexp, generator=g) * 0.2, torch.rand(cexp, generator=g) * 0.3 + 0.1
        w3 = (torch.randn((c, cexp), generator=g) / np.sqrt(cexp) * 0.6).to(tdt).float()
        b3 = torch.randn(c, generator=g) * 0.2
        pks.append(_rblock_packages(lib, code, c, cexp, w1, b1, s1, w2, b2, s2, w3, b3))
        e = F.prelu(F.conv2d(y, w1[:, :, None, None], b1), s1).to(tdt).float()
        d = F.prelu(F.conv2d(e, w2, b2, 1, 1, 1, cexp), s2).to(tdt).float()
        y = (F.conv2d(d, w3[:, :, None, None], b3) + y).to(tdt).float()
    ref = y.permute(0, 2, 3, 1).contiguous()
    pk = torch.cat(pks).to(device)
    per = pks[0].numel()
    xd = x.permute(0, 2, 3, 1).contiguous().to(tdt).to(device)
    # one launch per block (band configuration by batch size)
    cur, after = xd, []
    for i in range(nblk):
        nxt = torch.full_like(xd, float("nan"))
        _lib.check(lib.mfn_op_rblock(code, c, cexp, H, cur.data_ptr(), pk.data_ptr() + i * per, 1, nxt.data_ptr(), B,
                                     stream(device)), "rblock")
        cur = nxt
        after.append(nxt)
    # two blocks per launch, not in place: for a small batch of 14 x 14 maps these are two-row bands that compute their
    # neighbours' nearest rows themselves (blocks 0-1, then blocks 1-2 from the first block's output)
    for first, src in ((0, xd), (1, after[0])):
        two = torch.full_like(xd, float("nan"))
        _lib.check(lib.mfn_op_rblock(code, c, cexp, H, src.data_ptr(), pk.data_ptr() + first * per, 2, two.data_ptr(), B,
                                     stream(device)), "rblock, two blocks")
        torch.cuda.synchronize()
        if 7 * B <= torch.cuda.get_device_properties(device).multi_processor_count:
            # bands of a small batch run two slices of a block side by side and add the two halves of the project sum at
            # the end (kernels_rblock.hip, dual slices): the one-block launches' values to rounding, the same bits every time
            close(two, after[first + 1], 2 * rel, f"two blocks per launch, small batch, first block {first}")
            again = torch.full_like(xd, float("nan"))
            _lib.check(lib.mfn_op_rblock(code, c, cexp, H, src.data_ptr(), pk.data_ptr() + first * per, 2, again.data_ptr(), B,
                                         stream(device)), "rblock, two blocks, again")
            torch.cuda.synchronize()
            assert torch.equal(two, again), first
        else:
            assert torch.equal(two, after[first + 1]), first
    # one launch for all three, written over its own input
    chained = xd.clone()
    _lib.check(lib.mfn_op_rblock(code, c, cexp, H, chained.data_ptr(), pk.data_ptr(), nblk, chained.data_ptr(), B,
                                 stream(device)), "rblock chain")
    # the first images as a batch of their own (another configuration for the large batches)
    nb = min(B, 33)
    head = xd[:nb].clone()
    _lib.check(lib.mfn_op_rblock(code, c, cexp, H, head.data_ptr(), pk.data_ptr(), nblk, head.data_ptr(), nb,
                                 stream(device)), "rblock chain, small batch")
    torch.cuda.synchronize()
    assert torch.equal(cur, chained)
    assert torch.equal(head, chained[:nb])
    close(chained, ref, 6 * rel, f"rblock chain H={H} B={B}")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B", [1, 5, 37, 300])
def test_wide_wave_chain_matches_the_tile_per_wave_kernel(device, dt, B):
    """The chain of residual blocks on the wide-wave kernel (four waves per workgroup, one per SIMD, four image rows
    each: mfn_op_wblock) gives the bits of one mfn_op_rblock launch per block (bands for the small batches, whole-image
    workgroups of eight waves beyond), in place and for one, two and three blocks."""
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    c, cexp, H, nblk = 128, 256, 14, 3
    g = torch.Generator().manual_seed(11)
    pks = []
    for _ in range(nblk):
        w1 = (torch.randn((cexp, c), generator=g) / np.sqrt(c) * 1.3).to(tdt).float()
        w2 = (torch.randn((cexp, 1, 3, 3), generator=g) * 0.4).to(tdt).float()
        w3 = (torch.randn((c, cexp), generator=g) / np.sqrt(cexp) * 0.6).to(tdt).float()
        b1, b2, b3 = (torch.randn(n, generator=g) * 0.2 for n in (cexp, cexp, c))
        s1, s2 = (torch.rand(cexp, generator=g) * 0.3 + 0.1 for _ in range(2))
        pks.append(_rblock_packages(lib, code, c, cexp, w1, b1, s1, w2, b2, s2, w3, b3))
    pk = torch.cat(pks).to(device)
    per = pks[0].numel()
    x = torch.randn((B, H, H, c), generator=g).to(tdt).to(device)
    cur = x
    for n in range(1, nblk + 1):
        nxt = torch.full_like(x, float("nan"))
        _lib.check(lib.mfn_op_rblock(code, c, cexp, H, cur.data_ptr(), pk.data_ptr() + (n - 1) * per, 1, nxt.data_ptr(), B,
                                     stream(device)), "rblock")
        cur = nxt
        wide = x.clone()
        _lib.check(lib.mfn_op_wblock(code, c, cexp, H, wide.data_ptr(), pk.data_ptr(), n, wide.data_ptr(), B,
                                     stream(device)), "wblock")
        torch.cuda.synchronize()
        assert torch.equal(wide, cur), n
    assert lib.fra_device_fault(0, None) == 0


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("many", [False, True])
@pytest.mark.parametrize("cin,cexp,cout,H,stride,res,B", [(64, 128, 64, 56, 2, False, 2), (64, 128, 64, 28, 1, True, 3),
                                                         (64, 256, 128, 28, 2, False, 2), (128, 256, 128, 14, 1, True, 3),
                                                         (128, 512, 128, 14, 2, False, 2), (128, 256, 128, 7, 1, True, 5)])
def test_fused_block(device, dt, cin, cexp, cout, H, stride, res, B, many):
    """Fused expand/depthwise/project block against the three torch convs on 16-bit-rounded operands;
    the intermediate tensors are rounded to the storage type where the kernel rounds them.  The library runs
    a batch whose short bands fit the chip in one go (up to 18 or 36 images) with those and larger ones with
    tall bands: both are covered."""
    if many:
        B = 40
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    x = torch.randn((B, cin, H, H), generator=g).to(tdt).float()
    w1 = (torch.randn((cexp, cin), generator=g) / np.sqrt(cin) * 1.3).to(tdt).float()
    b1, s1 = torch.randn(cexp, generator=g) * 0.2, torch.rand(cexp, generator=g) * 0.3 + 0.1
    w2 = (torch.randn((cexp, 1, 3, 3), generator=g) * 0.4).to(tdt).float()
    b2, s2 = torch.randn(cexp, generator=g) * 0.2, torch.rand(cexp, generator=g) * 0.3 + 0.1
    w3 = (torch.randn((cout, cexp), generator=g) / np.sqrt(cexp)).to(tdt).float()
    b3 = torch.randn(cout, generator=g) * 0.2
    e = F.prelu(F.conv2d(x, w1[:, :, None, None], b1), s1).to(tdt).float()
    d = F.prelu(F.conv2d(e, w2, b2, stride, 1, 1, cexp), s2).to(tdt).float()
    y = F.conv2d(d, w3[:, :, None, None], b3)
    if res:
        y = y + x
    ref = y.permute(0, 2, 3, 1).contiguous()
    perm = torch.tensor([_perm(c) for c in range(cexp)])
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(cexp)                # inv[pos] = natural channel stored at pos
    dev = lambda t: t.contiguous().to(device)    # noqa: E731
    xd = dev(x.permute(0, 2, 3, 1).to(tdt))
    w1d, b1d, s1d = dev(w1.to(tdt)), dev(b1[inv]), dev(s1[inv] - 1.0)      # slopes travel as slope - 1
    w2d = dev(w2.reshape(cexp, 9).t()[:, inv].to(tdt))
    b2d, s2d = dev(b2[inv]), dev(s2[inv] - 1.0)
    w3d, b3d = dev(_w3_image(w3[:, inv], cout).to(tdt)), dev(b3)
    Ho = H // stride
    out = torch.full((B, Ho, Ho, cout), float("nan"), dtype=tdt, device=device)
    if (cin, cexp, stride) == (128, 256, 1) and H in (14, 7):
        # these shapes run in the register-resident residual kernel, which takes its weights as packages
        pk = _rblock_packages(lib, code, cin, cexp, w1, b1, s1, w2, b2, s2, w3, b3).to(device)
        _lib.check(lib.mfn_op_rblock(code, cin, cexp, H, xd.data_ptr(), pk.data_ptr(), 1, out.data_ptr(), B, stream(device)),
                   "rblock")
        torch.cuda.synchronize()
        close(out, ref, 3 * rel, f"rblock {cin}->{cexp}->{cout} H={H}")
        if H == 14:
            return
        out = torch.full((B, Ho, Ho, cout), float("nan"), dtype=tdt, device=device)   # the band kernel has this shape too
    _lib.check(lib.mfn_op_block(code, cin, cexp, cout, H, stride, int(res), xd.data_ptr(), w1d.data_ptr(),
                                b1d.data_ptr(), s1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), s2d.data_ptr(),
                                w3d.data_ptr(), b3d.data_ptr(), out.data_ptr(), B, stream(device)), "block")
    torch.cuda.synchronize()
    # intermediates are re-rounded, so allow a few output roundings
    close(out, ref, 3 * rel, f"block {cin}->{cexp}->{cout} H={H} s={stride}")


def test_pw_orientation_asymmetric(device):
    """A = identity against an asymmetric weight matrix exposes a transposed output or operand map."""
    lib = _lib.load()
    K = N = 64
    a = torch.eye(K, dtype=torch.float16)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 61 - 30).to(torch.float16)
    bias = torch.zeros(N)
    ad, wd, bd = a.to(device), w.to(device), bias.to(device)
    out = torch.empty((K, N), dtype=torch.float16, device=device)
    _lib.check(lib.mfn_op_pw(_lib.FRA_F16, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, None, out.data_ptr(), K,
                             K, N, stream(device)), "pw")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().float(), w.float().t())


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B", [1, 70])
def test_gdconv7(device, dt, B):
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    C_ = 512
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, C_, 7, 7), generator=g).to(tdt).float()
    w = (torch.randn((C_, 1, 7, 7), generator=g) * 0.15).to(tdt).float()
    bias = torch.randn(C_, generator=g) * 0.1
    ref = F.conv2d(x, w, bias, 1, 0, 1, C_).reshape(B, C_)
    xin = x.permute(0, 2, 3, 1).contiguous().to(tdt).to(device)
    wk = w.reshape(C_, 49).t().contiguous().to(tdt).to(device)
    out = torch.empty((B, C_), dtype=tdt, device=device)
    bd = bias.to(device)
    _lib.check(lib.mfn_op_gdconv7(code, xin.data_ptr(), wk.data_ptr(), bd.data_ptr(), out.data_ptr(), B, C_,
                                  stream(device)), "gdconv7")
    torch.cuda.synchronize()
    close(out, ref, rel, "gdconv7")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B", [1, 4, 7, 70])
def test_fused_tail(device, dt, B):
    """conv_6_sep (1x1 + PReLU) and conv_6_dw (7x7 global depthwise) in one kernel against the two torch convs."""
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(6)
    x = torch.randn((B, 128, 7, 7), generator=g).to(tdt).float()
    w6 = (torch.randn((512, 128), generator=g) / np.sqrt(128) * 1.3).to(tdt).float()
    b6, s6 = torch.randn(512, generator=g) * 0.2, torch.rand(512, generator=g) * 0.3 + 0.1
    w7 = (torch.randn((512, 1, 7, 7), generator=g) * 0.15).to(tdt).float()
    b7 = torch.randn(512, generator=g) * 0.1
    e = F.prelu(F.conv2d(x, w6[:, :, None, None], b6), s6)
    ref = F.conv2d(e, w7, b7, 1, 0, 1, 512).reshape(B, 512)
    dev = lambda t: t.contiguous().to(device)    # noqa: E731
    xd = dev(x.permute(0, 2, 3, 1).to(tdt))
    w6d, b6d, s6d = dev(w6.to(tdt)), dev(b6), dev(s6 - 1.0)
    w7d, b7d = dev(w7.reshape(512, 49).t().to(tdt)), dev(b7)
    out = torch.full((B, 512), float("nan"), dtype=tdt, device=device)
    _lib.check(lib.mfn_op_tail(code, xd.data_ptr(), w6d.data_ptr(), b6d.data_ptr(), s6d.data_ptr(), w7d.data_ptr(),
                               b7d.data_ptr(), out.data_ptr(), B, stream(device)), "tail")
    torch.cuda.synchronize()
    close(out, ref, 2 * rel, "fused tail")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B,E", [(1, 512), (33, 512), (64, 128), (5, 256), (40, 1024)])
def test_linear_l2norm(device, dt, B, E):
    tdt, code, rel = DT[dt]
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    a = torch.randn((B, 512), generator=g).to(tdt)
    w = (torch.randn((E, 512), generator=g) / 22.0).to(tdt)
    bias = torch.randn(E, generator=g) * 0.1
    y = a.float() @ w.float().t() + bias
    ref = y / y.norm(dim=1, keepdim=True)
    out = torch.full((B, E), float("nan"), dtype=torch.float32, device=device)
    ad, wd, bd = a.to(device), w.to(device), bias.to(device)   # keep the device copies alive
    _lib.check(lib.mfn_op_linear_l2norm(code, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, 512, E,
                                        stream(device)), "linear")
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=2e-6), (out.cpu() - ref).abs().max()
    assert torch.allclose(out.cpu().norm(dim=1), torch.ones(B), atol=1e-5)


@pytest.mark.parametrize("E", [128, 512, 1024])
def test_linear_l2norm_independent_of_batch(device, E):
    """A row's embedding must not depend on the batch it arrived in, to the last bit."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(6)
    a = torch.randn((40, 512), generator=g).half().to(device)
    w = (torch.randn((E, 512), generator=g) / 22.0).half().to(device)
    bias = (torch.randn(E, generator=g) * 0.1).to(device)
    outs = []
    for B in (40, 7):
        out = torch.full((B, E), float("nan"), dtype=torch.float32, device=device)
        _lib.check(lib.mfn_op_linear_l2norm(DT["f16"][1], a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, 512,
                                            E, stream(device)), "linear")
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0][:7], outs[1])


@pytest.mark.parametrize("B,E", [(1, 512), (7, 512), (130, 128), (3, 100)])
def test_tta_combine_and_l2norm(device, B, E):
    from face_recognition_amd.face_model import l2_norm, tta_combine
    g = torch.Generator().manual_seed(6)
    a, b = torch.randn((B, E), generator=g), torch.randn((B, E), generator=g)
    ref = (a + b) / (a + b).norm(dim=1, keepdim=True)
    out = tta_combine(a.to(device), b.to(device)).cpu()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-7)
    out1 = l2_norm(a.to(device)).cpu()
    assert torch.allclose(out1, a / a.norm(dim=1, keepdim=True), rtol=1e-5, atol=1e-7)


def test_l2norm_zero_row_is_nan(device):
    """No epsilon in the reference (face_model.py:19-22): 0/0 -> NaN, other rows unaffected."""
    from face_recognition_amd.face_model import l2_norm
    x = torch.ones((2, 512), device=device)
    x[0] = 0
    y = l2_norm(x).cpu()
    assert torch.isnan(y[0]).all() and torch.allclose(y[1].norm(), torch.tensor(1.0))


def test_bad_arguments_report_errors(device):
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.mfn_create(C.byref(h), 0, 500, _lib.FRA_F16) != 0
    assert b"embedding_size" in lib.fra_last_error()
    assert lib.mfn_create(C.byref(h), 0, 512, _lib.FRA_F32) != 0
    assert lib.mfn_create(C.byref(h), 99, 512, _lib.FRA_F16) != 0
    assert lib.mfn_create(C.byref(h), 0, 512, _lib.FRA_F16) == 0
    out = torch.empty(4, device=device)
    rc = lib.mfn_forward(h, out.data_ptr(), 1, out.data_ptr(), out.data_ptr(), 4, None)
    assert rc == 2 and b"before mfn_load_weights" in lib.fra_last_error()
    lib.mfn_destroy(h)
    assert lib.mfn_op_pw(_lib.FRA_F16, out.data_ptr(), out.data_ptr(), out.data_ptr(), None, None, out.data_ptr(), 10,
                         96, 64, None) != 0
    assert lib.mfn_op_dw3x3(_lib.FRA_F16, out.data_ptr(), out.data_ptr(), out.data_ptr(), None, out.data_ptr(), 1, 10,
                            10, 64, 1, None) != 0
    # one residual block in place: the bands of a small batch would read rows their neighbours have already replaced
    big = torch.zeros(4096, dtype=torch.float16, device=device)
    assert lib.mfn_op_rblock(_lib.FRA_F16, 128, 256, 7, big.data_ptr(), big.data_ptr(), 1, big.data_ptr(), 1, None) != 0
    assert b"in place" in lib.fra_last_error()
    assert lib.mfn_op_rblock(_lib.FRA_F16, 128, 256, 28, big.data_ptr(), big.data_ptr(), 2, big.data_ptr(), 1, None) != 0
    assert b"no kernel" in lib.fra_last_error()
