import sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd import _lib
lib = _lib.load(); device = "cuda:0"
D = 512; rng = np.random.default_rng(D); P = 1237
fl = (rng.standard_normal((P, D)) * 0.2 + 0.05).astype(np.float32)
fr = (fl * 0.5 + rng.standard_normal((P, D)) * 0.2).astype(np.float32)
folds = rng.integers(0, 10, P)
dl, dr = torch.from_numpy(fl).to(device), torch.from_numpy(fr).to(device)
mu = torch.empty(D, device=device); sc = torch.empty(P, device=device)
val = folds != 3
m = np.mean(np.concatenate((fl[val], fr[val]), 0), 0)[None]
print("m dtype", m.dtype)
cl, cr = fl - m, fr - m
nl = np.sqrt(np.sum(np.power(cl, 2), 1)); nr = np.sqrt(np.sum(np.power(cr, 2), 1))
yl = cl / np.expand_dims(nl, 1); yr = cr / np.expand_dims(nr, 1)
want = np.sum(np.power(yl - yr, 2), 1)
mask = torch.from_numpy(val.astype(np.uint8)).to(device)
_lib.check(lib.ev_fold_scores(dl.data_ptr(), dr.data_ptr(), mask.data_ptr(), int(val.sum()), P, D, 0, mu.data_ptr(), sc.data_ptr(), _lib.current_stream_ptr(device)), "x")
g_mu = mu.cpu().numpy(); g_l = dl.cpu().numpy(); g_s = sc.cpu().numpy()
print("mu mismatches", (g_mu != m[0]).sum(), np.abs(g_mu - m[0]).max())
print("yl mismatches", (g_l != yl).sum(), "of", yl.size, np.abs(g_l - yl).max())
print("score mismatches", (g_s != want).sum(), np.abs(g_s - want).max())
# isolate: use gpu mu on host
cl2 = fl - g_mu[None]
print("centre equal given gpu mu:", np.array_equal(cl2 / np.expand_dims(np.sqrt(np.sum(np.power(cl2, 2), 1)), 1), g_l))
bad_rows = np.where((g_l != yl).any(1))[0]
print("rows with mismatch", len(bad_rows), "of", P)
for i in bad_rows[:5]:
    found = None
    for d in range(-3, 4):
        n = np.float32(nl[i])
        for _ in range(abs(d)): n = np.nextafter(n, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
        if np.array_equal((cl[i] / n).astype(np.float32), g_l[i]): found = d
    print("row", i, "numpy norm", nl[i], "gpu norm offset in ulps:", found, " sumsq", np.sum(np.power(cl[i], 2)))
# does plain division match when norm matches?
good = np.where(~(g_l != yl).any(1))[0]
print("good rows", len(good))
