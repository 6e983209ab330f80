"""fb_search_l2_topk_noalloc (scratch in a static device array, no allocation) against fb_search_l2_topk,
through the C ABI: identical outputs on two calls in a row, the second smaller than the first, so that
whatever the first call left in the shared array must not leak into the second."""
import ctypes as C

import numpy as np
import pytest
import torch

from face_recognition_amd import _lib, synthetic

pytestmark = pytest.mark.gpu


def _create(lib, rows, store, device):
    h = C.c_void_p()
    _lib.check(lib.fb_create(C.byref(h), device.index, rows.data_ptr(), 1, rows.shape[0], rows.shape[1],
                             _lib.DTYPE_BY_NAME[store], 0), "fb_create")
    return h


def _search(lib, fn, h, q, k, device):
    dist = torch.full((q.shape[0], k), -7.0, dtype=torch.float32, device=device)
    idx = torch.full((q.shape[0], k), -7, dtype=torch.int32, device=device)
    _lib.check(getattr(lib, fn)(h, q.data_ptr(), q.shape[0], k, dist.data_ptr(), idx.data_ptr(),
                                _lib.current_stream_ptr(device)), fn)
    torch.cuda.synchronize(device)
    return dist.cpu(), idx.cpu()


@pytest.mark.parametrize("store,N", [("f16", 20000), ("f32", 1000)])
def test_noalloc_search_equals_allocating_search(device, store, N):
    lib = _lib.load()
    g = torch.from_numpy(synthetic.unit_rows(N, 512, seed=3)).to(device)
    h = _create(lib, g, store, device)
    try:
        # two calls in a row on the shared array: 200 queries at k = 5, then 37 queries at k = 3
        for Q, k, seed in ((200, 5, 4), (37, 3, 5)):
            qn, _ = synthetic.planted_queries(g.cpu().numpy(), Q, seed=seed)
            q = torch.from_numpy(qn).to(device)
            want_d, want_i = _search(lib, "fb_search_l2_topk", h, q, k, device)
            dist = torch.full((Q, k), -7.0, dtype=torch.float32, device=device)
            idx = torch.full((Q, k), -7, dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            # the library counts its own allocator calls (every hipMalloc / hipFree / hipHostMalloc it makes goes through
            # one wrapper): the count must stand still across the call
            before = lib.fra_dbg_alloc_calls()
            _lib.check(lib.fb_search_l2_topk_noalloc(h, q.data_ptr(), Q, k, dist.data_ptr(), idx.data_ptr(),
                                                     _lib.current_stream_ptr(device)), "fb_search_l2_topk_noalloc")
            torch.cuda.synchronize(device)
            assert lib.fra_dbg_alloc_calls() == before
            got_d, got_i = dist.cpu(), idx.cpu()
            assert torch.equal(got_i, want_i), (store, Q, k)
            assert torch.equal(got_d, want_d), (store, Q, k)
            assert (got_i >= 0).all()
    finally:
        lib.fb_destroy(h)


def test_noalloc_search_refuses_what_does_not_fit(device):
    lib = _lib.load()
    g = torch.from_numpy(synthetic.unit_rows(3000, 512, seed=3)).to(device)
    # searches that exceed the 8 MiB array whatever the GPU's compute-unit count (which sets the number of partial lists):
    # fp32: 1024 x 3000 distances = 12 MB; f16: the 16-bit copies of 9000 queries alone are 9.2 MB
    for store, Q in (("f32", 1024), ("f16", 9000)):
        q = torch.from_numpy(synthetic.unit_rows(Q, 512, seed=9)).to(device)
        h = _create(lib, g, store, device)
        try:
            dist = torch.empty((Q, 8), dtype=torch.float32, device=device)
            idx = torch.empty((Q, 8), dtype=torch.int32, device=device)
            before = lib.fra_dbg_alloc_calls()
            rc = lib.fb_search_l2_topk_noalloc(h, q.data_ptr(), Q, 8, dist.data_ptr(), idx.data_ptr(),
                                               _lib.current_stream_ptr(device))
            assert rc == 1 and b"static array holds" in lib.fra_last_error()
            assert lib.fra_dbg_alloc_calls() == before
            # the allocating entry point takes the same search
            _lib.check(lib.fb_search_l2_topk(h, q.data_ptr(), Q, 8, dist.data_ptr(), idx.data_ptr(),
                                             _lib.current_stream_ptr(device)), "fb_search_l2_topk")
            torch.cuda.synchronize(device)
        finally:
            lib.fb_destroy(h)
