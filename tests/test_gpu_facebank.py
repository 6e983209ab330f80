"""Facebank matching on the GPU against the oracle (the demos' own expressions / fp64 brute force)
and the golden results recorded on the facebank the reference ships."""
import os

import numpy as np
import pytest
import torch

from face_recognition_amd import synthetic
from face_recognition_amd.facebank import Gallery, match_facebank, merge_topk, topk_search
from oracle import matching_ref as mref

pytestmark = pytest.mark.gpu


def test_shipped_facebank_golden(device, golden_dir):
    z = np.load(os.path.join(golden_dir, "facebank_match.npz"))
    fb = torch.from_numpy(z["facebank"]).to(device)
    q = torch.from_numpy(z["queries"]).to(device)
    for th in (60, 50):
        mn, idx, sc = match_facebank(q, fb, threshold=th)
        assert idx.dtype == torch.int64 and mn.dtype == torch.float32
        assert np.array_equal(idx.cpu().numpy(), z[f"idx_th{th}"])
        assert np.allclose(mn.cpu().numpy(), z[f"min_th{th}"], rtol=1e-5, atol=2e-6)
        assert np.allclose(sc.cpu().numpy(), z[f"score_th{th}"], rtol=1e-5, atol=2e-4)
    names = z["names"]
    mn, idx, _ = match_facebank(fb, fb)
    assert [names[i + 1] for i in idx.cpu().tolist()] == list(names[1:])
    assert float(mn.abs().max()) == 0.0


@pytest.mark.parametrize("Q,N", [(32, 1000), (1, 5), (8, 77), (3, 1)])
def test_match_vs_reference_expression(device, Q, N):
    t = torch.from_numpy(synthetic.unit_rows(N, 512, seed=2))
    t[N // 2] *= 0.8                        # non-unit row: mean of several embeddings (facebank.py:68)
    q, truth = synthetic.planted_queries(t.numpy(), Q, seed=4, target_cos=0.75)
    q = torch.from_numpy(q)
    want_min, want_idx, want_score = mref.match_broadcast(q, t, 60.0)
    mn, idx, sc = match_facebank(q.to(device), t.to(device), 60)
    assert np.array_equal(idx.cpu().numpy(), want_idx.numpy())
    assert np.allclose(mn.cpu().numpy(), want_min.numpy(), rtol=1e-5, atol=2e-6)
    assert np.allclose(sc.cpu().numpy(), want_score.numpy(), rtol=1e-5, atol=3e-4)
    # given the same distances, the score line of cam_demo.py:105 (multiply, then add, then clamp) is exact
    assert torch.equal(sc.cpu(), torch.clamp(mn.cpu() * -80 + 156, 0, 100))


def test_exact_ties_pick_lowest_index(device):
    g = synthetic.unit_rows(20, 512, seed=12)
    g[13] = g[4]
    g[17] = g[4]
    q = g[[4, 13]] + 0.0
    for store in ("f32", "f16"):
        gal = Gallery(torch.from_numpy(g), store=store, device=device)
        d, i = gal.search(torch.from_numpy(q).to(device), 3)
        assert i.cpu().tolist() == [[4, 13, 17], [4, 13, 17]], store


def _check_topk(dist, idx, queries, gallery, k, tie_tol):
    """Top-1 identical to fp64 brute force except near-ties (|delta d2| < tie_tol); every returned
    distance is the true distance of the returned row within tolerance; lists ascending."""
    bd, bi = mref.l2_topk_bruteforce(queries, gallery, k)
    dist = dist.cpu().numpy().astype(np.float64)
    idx = idx.cpu().numpy().astype(np.int64)
    assert (np.diff(dist, axis=1) >= -1e-6).all()
    mism = idx[:, 0] != bi[:, 0]
    if mism.any():
        q64 = queries.astype(np.float64)
        d_mine = ((q64[mism] - gallery[idx[mism, 0]].astype(np.float64)) ** 2).sum(1)
        assert (np.abs(d_mine - bd[mism, 0]) < tie_tol).all(), "top-1 differs beyond the near-tie tolerance"
    true_d = ((queries.astype(np.float64)[:, None, :] - gallery[idx].astype(np.float64)) ** 2).sum(-1)
    assert np.abs(true_d - dist).max() < tie_tol
    # every returned row is at least as close as the true k-th neighbour, up to the near-tie tolerance
    assert (true_d <= bd[:, k - 1:k] + tie_tol).all()
    assert all(len(set(row)) == k for row in idx.tolist()), "duplicate rows in a top-k list"
    recall = np.mean([len(set(idx[i]) & set(bi[i])) / k for i in range(len(idx))])
    return float(mism.mean()), recall


@pytest.mark.parametrize("Q,N,k", [(300, 20000, 5), (1, 4097, 5), (17, 64, 8), (40, 130, 1), (5, 3, 5), (257, 70000, 5)])
def test_topk_f16_vs_bruteforce(device, Q, N, k):
    g = synthetic.unit_rows(N, 512, seed=3)
    q, truth = synthetic.planted_queries(g, Q, seed=4, target_cos=0.7)
    gal = Gallery(torch.from_numpy(g), store="f16", device=device)
    dist, idx = gal.search(torch.from_numpy(q).to(device), k)
    kk = min(k, N)
    if N < k:
        assert (idx[:, N:] == -1).all() and torch.isinf(dist[:, N:]).all()
    mism, recall = _check_topk(dist[:, :kk], idx[:, :kk], q, g, kk, 1e-3)
    planted = truth >= 0
    assert (idx.cpu().numpy()[planted, 0] == truth[planted]).all()
    print(f"Q={Q} N={N} k={k}: top-1 mismatch {mism:.4f}, top-{kk} recall {recall:.4f}")
    assert recall > 0.99   # fp16 rounding may swap the 5th and 6th neighbour when they are near-tied


def test_topk_f16_one_million_rows_vs_bruteforce(device):
    """The 10k x 1M top-5 configuration's gallery size: 256 queries (half planted, half random -- the random
    half is where near-ties live) against the float64 brute force over all 10^6 rows."""
    N, Q, k = 1_000_000, 256, 5
    g = synthetic.unit_rows(N, 512, seed=3)
    q, truth = synthetic.planted_queries(g, Q, seed=4, target_cos=0.7)
    gal = Gallery(None, n=N, d=512, store="f16", device=device)
    for lo in range(0, N, 250_000):                      # host rows go through the library's bounce buffer
        gal.fill(lo, torch.from_numpy(g[lo:lo + 250_000]))
    dist, idx = gal.search(torch.from_numpy(q).to(device), k)
    mism, recall = _check_topk(dist, idx, q, g, k, 1e-3)
    planted = truth >= 0
    assert (idx.cpu().numpy()[planted, 0] == truth[planted]).all()
    rnd = ~planted
    bd, bi = mref.l2_topk_bruteforce(q[rnd], g, k)
    rnd_recall = np.mean([len(set(a) & set(b)) / k for a, b in zip(idx.cpu().numpy()[rnd].tolist(), bi.tolist())])
    print(f"1M rows: top-1 mismatch {mism:.4f} (all inside the 1e-3 near-tie band), top-5 recall {recall:.4f}, "
          f"random half {rnd_recall:.4f}")
    assert recall > 0.98 and rnd_recall > 0.97
    gal.close()


def test_two_pass_search_keeps_ties_across_the_passes(device):
    """Galleries of some hundred thousand rows are searched in two passes: the first few thousand rows give every
    query a score floor, the rest is searched with lists that start at that floor.  A row of the first part that is
    DUPLICATED in the second part ties with its copy at distance 0; both must come back, lower index first, and
    the rest of the list must be the brute force's -- for queries whose k-th best sits in the first part
    (duplicates of several rows) as well."""
    N, Q, k = 300_000, 600, 5
    g = synthetic.unit_rows(N, 512, seed=21)
    g = g.astype(np.float16).astype(np.float32)          # queries below equal stored rows exactly
    src = 10 + 50 * np.arange(64)                        # inside the first pass (its rows end in the thousands)
    dst = 200_000 + 37 * np.arange(64)
    g[dst] = g[src]
    for j in range(8):                                   # four more copies of eight rows: all five best tie at 0
        g[250_000 + 5 * j + np.arange(1, 4)] = g[src[j]]
        g[2000 + j] = g[src[j]]
    q = synthetic.unit_rows(Q, 512, seed=22)
    q[:64] = g[src]
    gal = Gallery(torch.from_numpy(g), store="f16", device=device)
    dist, idx = gal.search(torch.from_numpy(q).to(device), k)
    d, i = dist.cpu().numpy(), idx.cpu().numpy()
    for j in range(8):
        want = sorted([int(src[j]), 2000 + j, int(dst[j])] + [250_000 + 5 * j + t for t in (1, 2, 3)])[:k]
        assert i[j].tolist() == want, (j, i[j], want)
        assert np.all(d[j] == d[j, 0])
    for j in range(8, 64):
        assert i[j, :2].tolist() == [int(src[j]), int(dst[j])], (j, i[j])
        assert d[j, 0] == d[j, 1]
    mism, recall = _check_topk(dist[64:], idx[64:], q[64:], g, k, 1e-3)
    assert recall > 0.98
    # every returned row is at least as close as the brute force's k-th, so nothing fell below the floor by mistake
    bd, bi = mref.l2_topk_bruteforce(q[:64], g, k)
    assert np.abs(d[:64].astype(np.float64) - bd).max() < 1e-3
    gal.close()


def test_gallery_cache_follows_the_tensor_object(device):
    """match_facebank / topk_search keep a device copy per facebank TENSOR OBJECT.  A facebank that is freed
    and replaced by another of the same shape usually gets the same address from the caching allocator
    (with _version 0 again): the new rows must be matched, not the old copy."""
    a = torch.from_numpy(synthetic.unit_rows(5, 512, seed=30)).to(device)
    b_host = torch.from_numpy(synthetic.unit_rows(5, 512, seed=31))
    q = b_host[[3]].to(device)
    _, ia, _ = match_facebank(a[[2]].clone(), a)
    assert int(ia[0]) == 2
    addr = a.data_ptr()
    del a
    b = b_host.to(device)
    reused = b.data_ptr() == addr
    mn, ib, _ = match_facebank(q, b)
    assert int(ib[0]) == 3 and float(mn[0]) < 1e-6, f"matched a stale copy (address reused: {reused})"
    d, i = topk_search(q, b, k=1, store="f32")
    assert int(i[0, 0]) == 3
    # in-place updates through torch move _version: the copy is rebuilt
    b[3] = b[0]
    mn2, _, _ = match_facebank(q, b)
    assert float(mn2[0]) > 0.5               # the row q was taken from is gone
    print("allocator handed the old address back:", reused)


def test_topk_f32_exact(device):
    g = synthetic.unit_rows(500, 512, seed=3) * np.linspace(0.7, 1.0, 500, dtype=np.float32)[:, None]
    q, _ = synthetic.planted_queries(g, 20, seed=4)
    d, i = topk_search(torch.from_numpy(q).to(device), torch.from_numpy(g).to(device), k=5, store="f32")
    bd, bi = mref.l2_topk_bruteforce(q, g, 5)
    assert i.dtype == torch.int64 and np.array_equal(i.cpu().numpy(), bi)
    assert np.allclose(d.cpu().numpy(), bd, rtol=1e-5, atol=2e-6)


def test_f32_gallery_other_width(device):
    g = synthetic.unit_rows(50, 128, seed=3)
    q = synthetic.unit_rows(4, 128, seed=5)
    d, i = topk_search(torch.from_numpy(q).to(device), torch.from_numpy(g).to(device), k=2)
    bd, bi = mref.l2_topk_bruteforce(q, g, 2)
    assert np.array_equal(i.cpu().numpy(), bi)


def test_sharded_search_equals_whole(device):
    """Row shards with index_base + merge == search of the whole gallery (what 8 GPUs do with an
    all-gather in between)."""
    N, Q, k, R = 30000, 64, 5, 4
    g = synthetic.unit_rows(N, 512, seed=3)
    q, _ = synthetic.planted_queries(g, Q, seed=4)
    qd = torch.from_numpy(q).to(device)
    whole = Gallery(torch.from_numpy(g), store="f16", device=device)
    wd, wi = whole.search(qd, k)
    bounds = [(r * N // R, (r + 1) * N // R) for r in range(R)]
    parts = [Gallery(torch.from_numpy(g[b:e]), store="f16", device=device, index_base=b).search(qd, k) for b, e in bounds]
    md, mi = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
    assert torch.equal(mi, wi)
    assert torch.allclose(md, wd, rtol=0, atol=1e-6)
    od, oi = mref.merge_topk(np.stack([p[0].cpu().numpy() for p in parts]), np.stack([p[1].cpu().numpy() for p in parts]), k)
    assert np.array_equal(mi.cpu().numpy(), oi)


def test_gallery_fill_piecewise(device):
    g = synthetic.unit_rows(1000, 512, seed=3)
    q = torch.from_numpy(synthetic.unit_rows(9, 512, seed=8)).to(device)
    a = Gallery(torch.from_numpy(g), store="f16", device=device).search(q, 5)
    gal = Gallery(None, n=1000, d=512, store="f16", device=device)
    gal.fill(0, torch.from_numpy(g[:300]).to(device))
    gal.fill(300, torch.from_numpy(g[300:]))           # host rows
    b = gal.search(q, 5)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


def test_fp32_search_in_query_chunks(device):
    """An fp32 gallery's Q x N distance matrix is produced in chunks of queries (256 MB of scratch at most):
    1100 x 70000 needs two chunks and must equal the brute force row for row."""
    g = synthetic.unit_rows(70000, 512, seed=3)
    q, _ = synthetic.planted_queries(g, 1100, seed=4)
    d, i = topk_search(torch.from_numpy(q).to(device), torch.from_numpy(g).to(device), k=3, store="f32")
    bd, bi = mref.l2_topk_bruteforce(q, g, 3)
    assert np.array_equal(i.cpu().numpy(), bi)
    assert np.allclose(d.cpu().numpy(), bd, rtol=1e-5, atol=3e-6)


def test_fill_is_ordered_on_the_current_stream(device):
    """Gallery.fill of device rows is queued on the current stream without waiting: rows written by kernels queued just
    before it, on a side stream, are the rows a search queued after it finds."""
    g = torch.from_numpy(synthetic.unit_rows(6000, 512, seed=3)).to(device)
    gal = Gallery(None, n=6000, d=512, store="f16", device=device)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        big = torch.randn((4096, 4096), device=device)
        for _ in range(6):
            big = big @ big * 1e-3                      # keeps the stream busy while the rows are produced behind it
        rows = g * 1.0 + big[0, 0] * 0.0                # rows depend on the slow work
        gal.fill(0, rows)
        d, i = gal.search(g[[5, 4321]], 1)
    side.synchronize()
    assert i.cpu().flatten().tolist() == [5, 4321] and float(d.abs().max()) < 1e-3


def test_host_fill_waits_for_searches_in_flight(device):
    """Host rows are written on a stream of the handle's own: a search queued before the fill on the caller's stream
    must have read the OLD rows (Gallery.fill lets that stream drain first)."""
    old = torch.from_numpy(synthetic.unit_rows(200000, 512, seed=3)).to(device)
    new = synthetic.unit_rows(200000, 512, seed=33)
    gal = Gallery(old, store="f16", device=device)
    q = old[::1000].contiguous()                         # 200 queries, each the copy of a row of the old gallery
    big = torch.randn((4096, 4096), device=device)
    for _ in range(4):
        big = big @ big * 1e-3                           # the search below is queued behind this
    d, i = gal.search(q, 1)
    gal.fill(0, torch.from_numpy(new))                   # replaces every row
    torch.cuda.synchronize()
    assert torch.equal(i.flatten().cpu(), torch.arange(0, 200000, 1000, dtype=torch.int32)) and float(d.abs().max()) < 1e-3
    d2, i2 = gal.search(torch.from_numpy(new[[7, 150001]]).to(device), 1)
    assert i2.flatten().cpu().tolist() == [7, 150001]


def test_facebank_errors(device):
    with pytest.raises(RuntimeError):
        Gallery(torch.zeros((4, 100)), store="f16", device=device)      # packed search needs D == 512
    gal = Gallery(torch.from_numpy(synthetic.unit_rows(8, 512)), store="f32", device=device)
    with pytest.raises(RuntimeError):
        gal.search(torch.zeros((2, 512), device=device), 9)
    with pytest.raises(RuntimeError):
        gal.search(torch.zeros((2, 100), device=device), 1)
