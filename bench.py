#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one MobileFaceNet forward over a batch of 1024 synthetic 112x112 crops per GPU
(BASELINE.json configs[1]); inputs are resident in HBM before the timed region; the embedding path
is data parallel with no collective, so N GPUs process N batches per step (weak scaling).
Rank 0 prints ONE JSON line.  Besides the headline faces/s it carries
  roofline      -- the forward's dominant kernel against the roofline that binds it (algorithmic bytes vs 8 TB/s,
                   or multiply-adds vs the dense MFMA peak), the other side reported beside it
  cpu_baseline  -- the CPU oracle (un-fused fp32 PyTorch graph) timed on this host's cores
  search        -- 1:N top-5 search, 10k queries against a row-sharded fp16 gallery
                   (per-shard top-5 -> one all-gather -> merge), queries/s and MFMA fraction
  kernels       -- per-kernel-class time, bytes and FLOPs of one forward
  latency_us    -- small-batch forward latency
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from face_recognition_amd import layers, synthetic  # noqa: E402
from face_recognition_amd import face_model  # noqa: E402
from face_recognition_amd.face_model import MobileFaceNet  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, datasheet
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA, datasheet


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=400.0,
                    help="un-timed forwards for this long BEFORE the warm-up steps, so that a short run of a fresh GPU is "
                         "timed at the clocks a long run settles at (0: none); reported as config.preheat")
    ap.add_argument("--batch", type=int, default=1024, help="crops per GPU per step")
    # f16 only: bf16 activations measure a per-embedding cosine of 0.999 against the fp32 reference, below the 0.9999 bar,
    # so nothing benchmarked here may use them (the library keeps act_dtype="bf16" as a documented, non-conforming option)
    ap.add_argument("--dtype", default="f16", choices=["f16"])
    ap.add_argument("--chunk", type=int, default=-1, help="batch slice size inside the library (-1 = tuned default)")
    ap.add_argument("--gallery-rows", type=int, default=100_000_000, help="rows of the WHOLE gallery (sharded over GPUs)")
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--search-steps", type=int, default=3)
    ap.add_argument("--unfused", action="store_true", help="one kernel per conv instead of one per block")
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--search-only", action="store_true", help="skip the forward timing (profiling aid)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def kernel_label(name, kind, dt, variant=0):
    """Name of the device kernel an op runs, as rocprofv3 prints it (leading template arguments included);
    variant: what the library says it launches for this batch (MobileFaceNet.op_variants)."""
    T = "_Float16" if dt == "f16" else "__bf16"
    if kind == "block":
        b = next(b for b in layers.BLOCKS if b.prefix == name)
        if variant == 1:
            return f"rblock_kernel<{T}, {b.cin}, {b.cexp}, {b.hin}>"
        if variant == 2:
            return f"pblock_kernel<{T}, {b.cin}, {b.cexp}, {b.cout}, {b.hin}>"
        return f"block_kernel<{T}, {b.cin}, {b.cexp}, {b.cout}, {b.hin}, {b.stride}>"
    for c in layers.all_convs():
        if c.prefix == name:
            if kind == "pw":
                bm = 64 if c.cin == 512 else 128
                nt = 2 if c.cout % 128 == 0 else 1
                return f"pw_gemm_kernel<{T}, {c.cin}, {bm}, {nt}>"
            if kind == "dw3x3":
                return f"dw3x3_kernel<{T}, {c.stride}>"
    return {"stemdw": f"stemdw_kernel<{T}>", "stem": f"stem_kernel<{T}>", "gdconv7": f"gdconv7_kernel<{T}>", "linear": f"linear_l2norm_kernel<{T}, 16, 1>",
            "tail": f"tail_kernel<{T}>"}[kind]


def per_kernel_profile(net, x, reps, dt):
    """Average per-op milliseconds (HIP events on the launch stream, inside the library), grouped by kernel."""
    acc = None
    for _ in range(reps):
        for _ in range(4):      # the timed call synchronises; un-timed forwards in front of it keep the clocks where the timed region had them
            net(x)
        _, ms, table = net.forward_timed(x)
        acc = ms if acc is None else acc + ms
    ms = acc / reps
    B = x.shape[0]
    plan = net.launch_plan(B)       # op -> the op whose launch covers it (a chain of blocks is one launch)
    variants = net.op_variants(B)
    groups = {}
    for i, ((name, kind, flops, nbytes, wbytes), t) in enumerate(zip(table, ms)):
        g = groups.setdefault(kernel_label(name, kind, dt, variants[i]), dict(kind=kind, launches=0, ops=0, ms=0.0, bytes=0.0, flops=0.0))
        g["ops"] += 1
        g["ms"] += float(t)
        g["flops"] += flops * B
        g["bytes"] += wbytes
        if plan[i] == i:            # only a launch's first op reads activations from HBM and only its last writes them;
            g["launches"] += 1      # the blocks of a chain have equal shapes, so one op's in + out is the launch's
            g["bytes"] += nbytes * B
    return groups, float(ms.sum())


MAX_CLOCK_GHZ = 2.4       # MI355X shader clock ceiling; kernels with matrix work hold 1.9 to 2.2 GHz


def incu_bounds(groups, B):
    """The in-CU floor of every kernel beside its two rooflines.  profiles/r*_incu.json (scripts/gpu_incu.sh +
    scripts/incu_profile.py, batch 1024) holds, per kernel and launch, the clocks its executed instructions need on the
    matrix pipe (per SIMD), the LDS array (per compute unit) and the vector ALU (per SIMD); the largest of the three at
    the chip's clock ceiling is a time no schedule of THOSE instructions can beat, and `frac` = that floor / this run's
    launch time says how close the kernel is to it.  `frac_at_profiled_clock` divides by the kernel's own clocks in the
    profiled run instead (the clock the chip really held).  Returns ({kernel label: dict}, source file)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_incu.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
        except (OSError, ValueError):
            continue
        if prof.get("batch") != B:
            continue
        out = {}
        for label, v in groups.items():
            key = "void fra::" + label[:-1]
            hit = [k for k in prof["kernels"] if k.startswith(key + ",") or k.startswith(key + ">")]
            if len(hit) != 1 or not v["launches"]:
                out[label] = None
                continue
            p = prof["kernels"][hit[0]]
            m, busy = p["model_clocks_per_launch"], p["counter_busy_clocks_per_launch"]
            us = {k.replace("_clocks", "_us"): round(c / (MAX_CLOCK_GHZ * 1e3), 2) for k, c in m.items()}
            top = max(m, key=m.get)
            launch_us = v["ms"] / v["launches"] * 1e3
            hbm_us = v["bytes"] / v["launches"] / (HBM_PEAK_GBS * 1e9) * 1e6
            out[label] = dict(us, bound=top.replace("_clocks", ""), floor_us=us[top.replace("_clocks", "_us")],
                              launch_us=round(launch_us, 2), frac=round(us[top.replace("_clocks", "_us")] / launch_us, 4),
                              frac_at_profiled_clock=round(m[top] / busy["kernel_clocks"], 4) if busy.get("kernel_clocks") else None,
                              hbm_floor_us=round(hbm_us, 2), above_hbm_floor=bool(us[top.replace("_clocks", "_us")] > hbm_us))
        return out, os.path.relpath(path, ROOT)
    return {label: None for label in groups}, None


def roofline_by_class(groups, B):
    """The two layer classes the forward is judged on, from the per-kernel event times.  Depthwise and pointwise
    convs run FUSED in the block kernels (and conv_6_sep + GDConv in the tail kernel), so both classes are carried by
    the same launches: the depthwise class is priced on those launches' HBM traffic, the pointwise class on the
    matrix-core work (1x1 convs + embedding head only; depthwise and stem multiply-adds are not counted)."""
    pw_flops_per_face = 2.0 * sum(c.cin * c.cout * c.hout * c.hout for c in layers.all_convs()
                                  if c.k == 1 and c.groups == 1)
    blk = [v for v in groups.values() if v["kind"] in ("block", "tail", "pw", "dw3x3", "gdconv7")]
    lin = [v for v in groups.values() if v["kind"] == "linear"]
    stem = [v for v in groups.values() if v["kind"] in ("stemdw", "stem")]

    def hbm(vs):
        ms, by = sum(v["ms"] for v in vs), sum(v["bytes"] for v in vs)
        gbs = by / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "ms_per_step": round(ms, 4), "algorithmic_bytes": round(by), "achieved": round(gbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}

    ms_mm = sum(v["ms"] for v in blk + lin)
    fl = pw_flops_per_face * B + sum(v["flops"] for v in lin)
    tf = fl / (ms_mm * 1e-3) / 1e12
    out = {"depthwise_3x3_and_gdconv": dict(hbm(blk), kernels="fused block kernels + tail kernel"),
           "pointwise_1x1_and_linear": {"bound": "mfma", "ms_per_step": round(ms_mm, 4), "flops": fl, "achieved": round(tf, 1),
                                        "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                                        "kernels": "fused block kernels + tail kernel + embedding head"},
           "stem": hbm(stem)}
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds):
    """The CPU oracle on this host: batches of 32 crops, fp32, un-fused graph, for about `seconds` on up to 16
    threads, then a short single-thread sample, then the reference's broadcast match (cam_demo.py:97-99) of
    32 embeddings against 1000 rows."""
    from oracle import matching_ref, mobilefacenet_ref
    threads = min(16, os.cpu_count() or 1)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(1, 512).items()}
    x = torch.from_numpy(synthetic.make_crops(32, 0))

    def timed(nthreads, budget, cap):
        torch.set_num_threads(nthreads)
        for _ in range(2):
            y = mobilefacenet_ref.forward(sd, x)
        n, t0 = 0, time.perf_counter()
        while True:
            y = mobilefacenet_ref.forward(sd, x)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= cap:
                return n, el, y

    n, el, y = timed(threads, seconds, 400)
    n1, el1, _ = timed(1, min(6.0, seconds / 2), 50)
    torch.set_num_threads(threads)
    bank = torch.from_numpy(synthetic.unit_rows(1000, 512, seed=2))
    for _ in range(3):
        matching_ref.match_broadcast(y, bank, 60)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        matching_ref.match_broadcast(y, bank, 60)
        ts.append(time.perf_counter() - t0)
    return {"value": round(32 * n / el, 1), "unit": "faces/s", "cores": threads, "kind": "port",
            "sample": f"{n} batches of 32 crops (fp32 un-fused PyTorch graph of the oracle) in {el:.1f} s; "
                      f"host has {os.cpu_count()} logical CPUs",
            "one_thread_faces_per_s": round(32 * n1 / el1, 1),
            "match_32x1000_queries_per_s": round(32 / float(np.median(ts)), 1),
            "cpu_model": _cpu_model(), "torch": torch.__version__}


CHUNK_ROWS = 1 << 18  # the synthetic gallery is defined in global chunks of 262,144 rows


def gallery_chunk(c, dev):
    """Rows [c*CHUNK_ROWS, (c+1)*CHUNK_ROWS) of the logical gallery: unit-norm N(0,I), a function of c only,
    so every sharding of the gallery holds the same rows."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234567 + c)
    r = torch.randn((CHUNK_ROWS, 512), generator=g, device=dev, dtype=torch.float32)
    return r / r.norm(dim=1, keepdim=True)


def sampled_bruteforce(q, gal_chunks, total, k, sample, dev):
    """fp32 brute force (torch matmul on the device, chunk by chunk) for every `sample`-th query: the check beside
    a search timing.  Returns (top-1 agreement helper tensors) idx [n,k] int64 and dist [n,k] of the exact top-k."""
    qs = q[::sample]
    best_d = torch.full((qs.shape[0], k), float("inf"), device=dev)
    best_i = torch.full((qs.shape[0], k), -1, dtype=torch.int64, device=dev)
    qn = (qs * qs).sum(1, keepdim=True)
    for c in range((total + CHUNK_ROWS - 1) // CHUNK_ROWS):
        rows = gal_chunks(c, dev)[: min(CHUNK_ROWS, total - c * CHUNK_ROWS)].half().float()   # the stored (fp16) values
        d = qn + (rows * rows).sum(1)[None, :] - 2.0 * (qs @ rows.t())
        cd, ci = torch.topk(d, k, dim=1, largest=False)
        alld = torch.cat([best_d, cd], 1)
        alli = torch.cat([best_i, ci + c * CHUNK_ROWS], 1)
        o = torch.argsort(alld, dim=1, stable=True)[:, :k]
        best_d, best_i = torch.gather(alld, 1, o), torch.gather(alli, 1, o)
    return best_d, best_i


def run_search(args, rank, world, dev, dist_mod, total=None, check_sample=0):
    from face_recognition_amd.facebank import Gallery
    from face_recognition_amd.sharded import ShardedGallery, result_digest, shard_bounds
    total = args.gallery_rows if total is None else total
    b, e = shard_bounds(total, world, rank)
    gal = Gallery(None, n=e - b, d=512, store="f16", device=dev, index_base=b)
    t0 = time.perf_counter()
    c0, c1 = b // CHUNK_ROWS, (e - 1) // CHUNK_ROWS
    for c in range(c0, c1 + 1):
        rows = gallery_chunk(c, dev)
        lo, hi = max(b, c * CHUNK_ROWS), min(e, (c + 1) * CHUNK_ROWS)
        gal.fill(lo - b, rows[lo - c * CHUNK_ROWS: hi - c * CHUNK_ROWS])
    del rows
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    # queries: half planted near rows of chunk 0 (true nearest row known), half random; same on every rank
    Q = args.queries
    first = gallery_chunk(0, dev)[: min(CHUNK_ROWS, total)]
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    pick = torch.randint(0, first.shape[0], (Q // 2,), generator=g, device=dev)
    noise = torch.randn((Q // 2, 512), generator=g, device=dev)
    noise = noise / noise.norm(dim=1, keepdim=True)
    planted = first[pick] + float(np.sqrt(1 / 0.7 ** 2 - 1)) * noise
    rnd = torch.randn((Q - Q // 2, 512), generator=g, device=dev)
    q = torch.cat([planted, rnd])
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    del first
    sg = ShardedGallery(total, gal.search)
    d, i = sg.search(q, 5)          # warm-up + correctness of the planted half
    torch.cuda.synchronize()
    hit = float((i[: Q // 2, 0].to(torch.int64) == pick).float().mean())
    check = None
    if check_sample and world == 1:
        # every check_sample-th query (planted and random halves alike) against an fp32 brute force over all rows
        bd, bi = sampled_bruteforce(q, gallery_chunk, total, 5, check_sample, dev)
        mine_i, mine_d = i[::check_sample].to(torch.int64), d[::check_sample]
        same = mine_i[:, 0] == bi[:, 0]
        near = (mine_d[:, 0] - bd[:, 0]).abs() < 1e-3          # a different row at (nearly) the same distance
        recall = float(np.mean([len(set(a) & set(b)) / 5.0 for a, b in zip(mine_i.tolist(), bi.tolist())]))
        check = {"queries_checked": int(same.numel()), "top1_identical": round(float(same.float().mean()), 5),
                 "top1_identical_or_near_tie_1e-3": round(float((same | near).float().mean()), 5),
                 "top5_recall": round(recall, 5), "against": "fp32 brute force over all rows (torch, on the device)"}
        del bd, bi
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.search_steps):
        d, i = sg.search(q, 5)
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev if world == 1 or dist_mod.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist_mod.all_reduce(el, op=dist_mod.ReduceOp.MAX)
    el = float(el) / args.search_steps
    flops = 2.0 * 512 * Q * total
    # the three steps apart (a synchronisation after each, so their sum exceeds ms_per_batch), slowest rank per step
    parts = None
    for _ in range(2):
        d, i, parts = sg.search_timed(q, 5)
    pt = torch.tensor([parts["local_search_ms"], parts["allgather_ms"], parts["merge_ms"]], dtype=torch.float64,
                      device=dev if world == 1 or dist_mod.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist_mod.all_reduce(pt, op=dist_mod.ReduceOp.MAX)
    shard = {"rows_per_gpu": e - b, "local_search_ms": round(float(pt[0]), 3), "allgather_ms": round(float(pt[1]), 3),
             "merge_ms": round(float(pt[2]), 3), "lists_merged": world,
             "allgather_bytes_received_per_rank": (world - 1) * Q * 5 * 8}
    digest = result_digest(i)
    gal.close()
    return {"metric": "1:N top-5 queries/sec", "value": round(Q / el, 1), "unit": "queries/s", "queries": Q,
            "gallery_rows": total, "rows_per_gpu": e - b, "k": 5, "gallery_dtype": "f16", "ms_per_batch": round(el * 1e3, 3),
            "scaling": "strong", "planted_top1_hit_rate": round(hit, 5), "bruteforce_check": check,
            # SHA-256 of the [Q,5] int32 index tensor (16 hex digits): the same gallery and queries give the same digest
            # at every world size, so an N-GPU run is checked against the 1-GPU run by comparing this string
            "result_digest": digest, "shard": shard,
            "gallery_build_s": round(build_s, 2),
            "merge": (("RCCL" if dist_mod.get_backend() == "nccl" else dist_mod.get_backend()) + " all_gather + fb_merge_topk") if world > 1 else "none (1 shard)",
            "roofline": {"bound": "mfma", "achieved": round(flops / el / 1e12, 2), "peak": MFMA_PEAK_TFLOPS * world,
                         "unit": "TFLOP/s", "frac": round(flops / el / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
                         "traffic": None}}


def shard_rehearsal(args, dev, shards=8):
    """What ONE rank of an `shards`-GPU run of the sharded search does, timed on this GPU: the search of shard 0 (rows
    [0, total / shards) of the same logical gallery, same queries) and the merge of `shards` candidate lists.  The
    all-gather between them cannot be measured on one GPU: (shards - 1) * Q * 5 * 8 bytes arrive per rank over xGMI
    (2.8 MB at 8 ranks and 10k queries; some 20 us at 153 GB/s per link) -- 0.05 ms is assumed for it."""
    from face_recognition_amd.facebank import Gallery, merge_topk
    from face_recognition_amd.sharded import shard_bounds
    total, Q = args.gallery_rows, args.queries
    b, e = shard_bounds(total, shards, 0)
    gal = Gallery(None, n=e - b, d=512, store="f16", device=dev, index_base=b)
    for c in range(b // CHUNK_ROWS, (e - 1) // CHUNK_ROWS + 1):
        rows = gallery_chunk(c, dev)
        lo, hi = max(b, c * CHUNK_ROWS), min(e, (c + 1) * CHUNK_ROWS)
        gal.fill(lo - b, rows[lo - c * CHUNK_ROWS: hi - c * CHUNK_ROWS])
    del rows
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    q = torch.randn((Q, 512), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    d, i = gal.search(q, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.search_steps):
        d, i = gal.search(q, 5)
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t0) / args.search_steps * 1e3
    # `shards` lists per query: this shard's own and copies of it with the indices of the other shards' rows
    gd = torch.stack([d + 8.0 * r for r in range(shards)])          # squared distances of unit vectors stay below 4
    gi = torch.stack([i + r * (e - b) for r in range(shards)]).to(torch.int32)
    merge_topk(gd, gi, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        md, mi = merge_topk(gd, gi, 5)
    torch.cuda.synchronize()
    merge_ms = (time.perf_counter() - t0) / 20 * 1e3
    assert torch.equal(mi.to(torch.int64), i.to(torch.int64))          # the unshifted list wins every slot
    gal.close()
    assumed = 0.05
    return {"shards": shards, "rows_per_gpu": e - b, "local_search_ms": round(local_ms, 3), "merge_ms": round(merge_ms, 3),
            "allgather_assumed_ms": assumed, "predicted_ms": round(local_ms + merge_ms + assumed, 3)}


def latency(net, dev):
    """Small-batch latency (BASELINE configs[2]): host wall time per call incl. the final sync, eager launches
    vs one hipGraph replay of forward + match against a 1000-row facebank."""
    from face_recognition_amd.facebank import match_facebank
    from face_recognition_amd.graphed import GraphedRecognizer
    bank = torch.from_numpy(synthetic.unit_rows(1000, 512, seed=2)).to(dev)
    rec = GraphedRecognizer(net, bank, tta=False, threshold=60)
    rec_tta = GraphedRecognizer(net, bank, tta=True, threshold=60)
    bank5 = torch.from_numpy(synthetic.unit_rows(5, 512, seed=4)).to(dev)      # the size of the facebank the reference ships
    rec5 = GraphedRecognizer(net, bank5, tta=False, threshold=60)

    def timeit(fn, x, reps=1000, warm=100):
        for _ in range(warm):
            fn(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts)) * 1e6
        return {"p50": round(float(ts[len(ts) // 2]), 1), "p99": round(float(ts[int(len(ts) * 0.99)]), 1)}

    out = {}
    for B in (1, 2, 4, 8):
        x = torch.rand((B, 3, 112, 112), device=dev) * 2 - 1
        out[f"b{B}"] = {"eager_forward": timeit(net, x, 300, 30),
                        "eager_forward_match": timeit(lambda t: match_facebank(net(t), bank, 60), x, 300, 30),
                        "graph_forward_match": timeit(rec, x),
                        # the crops written straight into the graph's input (a per-frame loop's H2D copy lands there)
                        "graph_forward_match_static_input": timeit(rec, rec.input_buffer(B).copy_(x)),
                        "graph_forward_match_bank5": timeit(rec5, x),
                        "graph_tta_forward_match": timeit(rec_tta, x)}
    return out


def verification_timing(dev):
    """10-fold protocol on 6000 synthetic pairs: features on the device vs the host numpy path (same results)."""
    from face_recognition_amd import Evaluation
    fl, fr, ds = synthetic.planted_features(6000, seed=21, same_cos=0.5, diff_cos=0.12)
    dl, dr = torch.from_numpy(fl).to(dev), torch.from_numpy(fr).to(dev)
    Evaluation.evaluation_10_fold(dl, dr, ds, "cos_distance")            # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter(); a_dev, t_dev = Evaluation.evaluation_10_fold(dl, dr, ds, "cos_distance")
    t1 = time.perf_counter(); a_host, t_host = Evaluation.evaluation_10_fold(fl, fr, ds, "cos_distance")
    t2 = time.perf_counter()
    return {"pairs": 6000, "method": "cos_distance", "device_ms": round((t1 - t0) * 1e3, 1),
            "host_numpy_ms": round((t2 - t1) * 1e3, 1),
            "identical": bool(np.array_equal(a_dev, a_host) and np.array_equal(t_dev, t_host)),
            "accuracy": round(float(a_dev.mean()) * 100, 3)}


def verification_12k(net, dev, pairs=12000, batch=256):
    """BASELINE configs[1], second half: 12k synthetic verification pairs through getFeature (flip-TTA: four
    crop sets per loader batch; getFeature runs consecutive loader batches of up to 1024 crops together as one
    forward -- rows do not depend on the batch they are computed in) and the 10-fold protocol on the device.  Same-pairs are two
    noisy copies of one template, different-pairs two templates; crops are resident in HBM, as `value`'s are."""
    from face_recognition_amd import Evaluation
    ds = synthetic.PairSet(pairs)
    flags = torch.tensor(ds.flags, device=dev).view(-1, 1, 1, 1)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    up = lambda t: t.repeat_interleave(8, 2).repeat_interleave(8, 3)    # noqa: E731
    t1 = up(torch.rand((pairs, 3, 14, 14), generator=g, device=dev) * 2 - 1)
    t2 = up(torch.rand((pairs, 3, 14, 14), generator=g, device=dev) * 2 - 1)
    left = t1 + 0.3 * torch.randn(t1.shape, generator=g, device=dev)
    right = torch.where(flags == 1, t1, t2) + 0.3 * torch.randn(t1.shape, generator=g, device=dev)
    del t1, t2
    lf, rf = torch.flip(left, dims=[3]), torch.flip(right, dims=[3])
    loader = [[left[i:i + batch], lf[i:i + batch], right[i:i + batch], rf[i:i + batch]] for i in range(0, pairs, batch)]
    Evaluation.getFeature(net, loader[:2], dev, flip=True, keep_on_device=True)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fl, fr = Evaluation.getFeature(net, loader, dev, flip=True, keep_on_device=True)
    torch.cuda.synchronize()
    t1_ = time.perf_counter()
    accs, thr = Evaluation.evaluation_10_fold(fl, fr, ds, "cos_distance")
    t2_ = time.perf_counter()
    return {"pairs": pairs, "loader_batch": batch, "crops_per_forward": max(4 * batch, 1024 // (4 * batch) * 4 * batch),
            "crops_embedded": 4 * pairs, "embed_s": round(t1_ - t0, 4),
            "faces_per_s": round(4 * pairs / (t1_ - t0), 1), "protocol_10fold_ms": round((t2_ - t1_) * 1e3, 1),
            "accuracy": round(float(np.mean(accs)) * 100, 3), "threshold": round(float(np.mean(thr)), 4)}


def main():
    args = parse()
    if os.environ.get("FRA_BENCH_WATCHDOG"):     # dump every thread's stack and exit if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FRA_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # rehearsal on a one-GPU box: FRA_BENCH_ONE_GPU=1 puts every rank on cuda:0, FRA_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device); the control flow and the collectives' shapes are the real ones
    if os.environ.get("FRA_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FRA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    sd = synthetic.make_state_dict(1, 512)
    net = MobileFaceNet(512, act_dtype=args.dtype)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net = net.to(dev).eval()
    if args.chunk >= 0:
        net.set_batch_chunk(args.chunk)
    if args.unfused:
        net.set_fused(False)
    B = args.batch
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    x = (torch.randint(0, 256, (B, 3, 112, 112), generator=g, device=dev).float() / 255.0 - 0.5) / 0.5

    if args.search_only:
        # the full-size gallery: every 500th query (10 planted, 10 random) against the fp32 brute force over all rows
        s = run_search(args, rank, world, dev, dist, check_sample=500)
        if rank == 0:
            print(json.dumps(s), flush=True)
        return
    # a GPU that has been idle raises its clocks over the first few hundred milliseconds of load: five warm-up steps are
    # 6 ms.  The same forwards, un-timed and counted, until the clock has had that time
    preheat_steps, t_pre = 0, time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(8):
            y = net(x)
        torch.cuda.synchronize()
        preheat_steps += 8
    for _ in range(args.warmup):
        y = net(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = net(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if world == 1 or dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el)
    # timing-only diagnostic builds (scripts/build_alt.sh) compute garbage on purpose and set this
    if not os.environ.get("FRA_BENCH_UNCHECKED"):
        assert bool(torch.isfinite(y).all()) and abs(float(y[0].norm()) - 1.0) < 1e-3
    faces_per_s = world * B * args.steps / el

    result = {
        "metric": "aligned faces/sec embedded (MobileFaceNet forward)", "value": round(faces_per_s, 1), "unit": "faces/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "MobileFaceNet(512) forward, 1024 synthetic 112x112 crops per GPU per step, seeded "
                               "random-init weights with non-trivial BatchNorm/PReLU (BASELINE configs[1]); "
                               "data-parallel embedding, no collective",
                   "batch_per_gpu": B, "global_batch": B * world, "input": "f32 NCHW resident in HBM",
                   "output": "f32 [B,512] L2-normalised", "parallelism": f"dp{world}",
                   "batch_chunk": net._batch_chunk, "fused_blocks": net._fused,
                   "preheat": {"ms": args.preheat_ms, "untimed_steps_before_warmup": preheat_steps},
                   # 1: batches of up to ~36 crops (the latency lines) add a block's project sum in two halves; 0: batch-invariant bits
                   "small_batch_mode": face_model.small_batch_mode()},
        "flops_per_face": 2 * layers.macs_per_face(512),
        "forward_tflops": round(faces_per_s * 2 * layers.macs_per_face(512) / 1e12, 2),
    }

    if rank == 0 and not args.no_latency:
        # the optional call shapes of the same forward (not `value`): raw uint8 HWC crops, and flip-TTA embeddings
        xu8 = torch.randint(0, 256, (B, 112, 112, 3), generator=g, device=dev, dtype=torch.uint8)
        extra = {}
        for name, fn in (("u8_hwc_input", lambda: net.forward_u8(xu8)), ("u8_hwc_input_flip_tta", lambda: net.forward_u8(xu8, tta=True))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            extra[name] = {"faces_per_s": round(B * args.steps / (time.perf_counter() - t1), 1)}
        # crops that start in pinned HOST memory (PCIe-inclusive, never `value`): copy + forward one after the other on
        # one stream, and through EmbeddingStream, which copies batch i + 1 while batch i runs
        from face_recognition_amd.streaming import EmbeddingStream
        hosts = [torch.randint(0, 256, (B, 112, 112, 3), generator=torch.Generator().manual_seed(5 + i), dtype=torch.uint8).pin_memory()
                 for i in range(3)]
        es = EmbeddingStream(net, max_batch=B, depth=3)

        def serial():
            for i in range(args.steps):
                net.forward_u8(hosts[i % 3].to(dev, non_blocking=True))

        def overlapped():
            for _ in es.embed(hosts[i % 3] for i in range(args.steps)):
                pass
        for name, fn in (("u8_hwc_pinned_host_serial", serial), ("u8_hwc_pinned_host_overlapped", overlapped)):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            extra[name] = {"faces_per_s": round(B * args.steps / (time.perf_counter() - t1), 1),
                           "host_bytes_per_face": 112 * 112 * 3}
        del es, hosts
        result["other_call_shapes"] = extra
    if rank == 0:
        groups, total_ms = per_kernel_profile(net, x, 10, args.dtype)
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        # HBM bytes per launch come from rocprofv3 PMC passes of this command (separate runs, scripts/gpu_traffic.sh):
        # the newest committed summary that knows the kernel, named in traffic_source; null when none does
        traffic, traffic_source = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(path) as f:
                    t = json.load(f)["kernels"].get(dom_name, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError, KeyError):
                continue
            if t is not None:
                traffic, traffic_source = t, os.path.relpath(path, ROOT)
                break
        # the roofline that binds the dominant kernel: whichever of (algorithmic bytes at the HBM peak, multiply-adds at
        # the dense MFMA peak) needs more time; the other side is reported next to it
        tfs = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        hbm_bound = dom["bytes"] / (HBM_PEAK_GBS * 1e9) >= dom["flops"] / (MFMA_PEAK_TFLOPS * 1e12)
        result["roofline"] = {"bound": "hbm" if hbm_bound else "mfma",
                              "achieved": round(gbs, 1) if hbm_bound else round(tfs, 1),
                              "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS,
                              "unit": "GB/s" if hbm_bound else "TFLOP/s",
                              "frac": round(gbs / HBM_PEAK_GBS, 4) if hbm_bound else round(tfs / MFMA_PEAK_TFLOPS, 4),
                              "traffic": traffic, "traffic_source": traffic_source,
                              "kernel": dom_name,
                              "hbm_gbps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                              "mfma_tflops": round(tfs, 1), "mfma_frac": round(tfs / MFMA_PEAK_TFLOPS, 4),
                              "launches_per_step": dom["launches"],
                              "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                              "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                              "share_of_forward": round(dom["ms"] / total_ms, 3)}
        incu, incu_source = incu_bounds(groups, B)
        result["kernels"] = {k: {"kind": v["kind"], "launches": v["launches"], "ms": round(v["ms"], 4),
                                 "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                 "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 "incu": incu[k]} for k, v in groups.items()}
        # the executed-instruction counts behind `incu` come from a committed profile (like roofline.traffic), the launch
        # times from this run; null where no profile of this batch size knows the kernel
        result["incu_source"] = incu_source
        if incu.get(dom_name):
            result["roofline"]["incu"] = incu[dom_name]
        result["sum_kernel_ms"] = round(total_ms, 4)
        result["roofline_by_class"] = roofline_by_class(groups, B)
        if not args.no_latency:
            result["latency_us"] = latency(net, dev)
            result["verification_10fold"] = verification_timing(dev)
            result["verification_12k_pairs"] = verification_12k(net, dev)
            # the reference harness's own batch size: 50 pairs = 200 crops per forward (test_lfw_agedb_cfpfp.py:60)
            result["verification_12k_pairs_bs50"] = verification_12k(net, dev, batch=50)
    if not args.no_search:
        if world == 1:      # BASELINE configs[3]: 10k queries x 1M fp16 rows, top-5, one GPU
            result["search_1m"] = run_search(args, rank, world, dev, dist, total=1_000_000, check_sample=40)
        # the full-size gallery: every 500th query (10 planted, 10 random) against the fp32 brute force over all rows
        s = run_search(args, rank, world, dev, dist, check_sample=500)
        if world == 1 and args.gallery_rows >= 8 * CHUNK_ROWS:
            # one rank's share of the 8-GPU run, timed here: what the 1 -> 8 GPU scaling of this search should come to
            s["predicted_8gpu"] = shard_rehearsal(args, dev, 8)
            s["predicted_8gpu_ms"] = s["predicted_8gpu"]["predicted_ms"]
            s["predicted_scaling_1_to_8"] = round(s["ms_per_batch"] / s["predicted_8gpu_ms"], 2)
        if rank == 0:
            result["search"] = s
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
