"""Prints the per-kernel table (event and rocprofv3 durations, algorithmic and measured bytes, GB/s, TFLOP/s) from
profiles/<tag>_bench.json, <tag>_kernel_stats.csv and <tag>_pmc_traffic.json:  python scripts/kernel_table.py [tag, default r02]"""
import csv, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = json.load(open(os.path.join(root, f"profiles/{tag}_bench.json")))
tr = json.load(open(os.path.join(root, f"profiles/{tag}_pmc_traffic.json")))["kernels"]
st = {}
for row in csv.DictReader(open(os.path.join(root, f"profiles/{tag}_kernel_stats.csv"))):
    n = row["Name"]
    mr = re.search(r"rblock_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n)
    mp = re.search(r"pblock_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n)
    if mp:
        st["pblock_kernel<_Float16, %s, %s, %s, %s>" % mp.groups()] = float(row["AverageNs"]) / 1e3
        continue
    m = None if mr else re.search(r"block_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n)
    if mr and mr.group(3) == mr.group(4):        # whole-image workgroups: the chained launch a full batch runs
        st["rblock_kernel<_Float16, %s, %s, %s>" % mr.groups()[:3]] = float(row["AverageNs"]) / 1e3
    elif m and int(m.group(6)) in (4, 7, 14):      # the tall-band configurations a full batch runs
        st["block_kernel<_Float16, %s, %s, %s, %s, %s>" % m.groups()[:5]] = float(row["AverageNs"]) / 1e3
    elif "stemdw_kernelIDF16_" in n: st["stemdw_kernel<_Float16>"] = float(row["AverageNs"]) / 1e3
    elif "tail_kernelIDF16_" in n: st["tail_kernel<_Float16>"] = float(row["AverageNs"]) / 1e3
    elif "linear_l2norm_kernelIDF16_Li16ELi1E" in n or "l2norm_rows_kernel" in n:      # the head is two launches: one row
        st["linear_l2norm_kernel<_Float16, 16, 1>"] = st.get("linear_l2norm_kernel<_Float16, 16, 1>", 0.0) + float(row["AverageNs"]) / 1e3
names = {"block_kernel<_Float16, 64, 128, 64, 28, 1>": "`block_kernel` 64→128→64 @28², s1 (conv_3, ×4)",
         "pblock_kernel<_Float16, 64, 128, 64, 28>": "`pblock_kernel` 64→128→64 @28², s1 (conv_3, ×4: tile pipeline)",
         "block_kernel<_Float16, 128, 256, 128, 14, 1>": "`block_kernel` 128→256→128 @14², s1 (conv_4, ×6)",
         "rblock_kernel<_Float16, 128, 256, 14>": "`rblock_kernel` 128→256→128 @14², s1 (conv_4: six blocks, ONE launch)",
         "block_kernel<_Float16, 64, 128, 64, 56, 2>": "`block_kernel` 64→128→64 @56², s2 (conv_23, ×1)",
         "stemdw_kernel<_Float16>": "`stemdw_kernel` (conv1 + conv2_dw, ×1)",
         "block_kernel<_Float16, 64, 256, 128, 28, 2>": "`block_kernel` 64→256→128 @28², s2 (conv_34, ×1)",
         "block_kernel<_Float16, 128, 512, 128, 14, 2>": "`block_kernel` 128→512→128 @14², s2 (conv_45, ×1)",
         "block_kernel<_Float16, 128, 256, 128, 7, 1>": "`block_kernel` 128→256→128 @7², s1 (conv_5, ×2)",
         "rblock_kernel<_Float16, 128, 256, 7>": "`rblock_kernel` 128→256→128 @7², s1 (conv_5: two blocks, ONE launch, four stacked images per workgroup)",
         "linear_l2norm_kernel<_Float16, 16, 1>": "`linear_l2norm_kernel` + `l2norm_rows_kernel`",
         "tail_kernel<_Float16>": "`tail_kernel` (conv_6_sep + conv_6_dw)"}
tb = tt = ts = 0.0
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
    alg = v["GBps"] * v["ms"] * 1e-3 * 1e9 / v["launches"] / 1e6
    pm = tr.get(k, {}).get("hbm_bytes_per_launch", float("nan")) / 1e6
    tb += alg * v["launches"]; tt += pm * v["launches"]; ts += st[k] * v["launches"]
    print("| %s | %.3f | %.1f | %.1f | %.0f | %.0f | %.0f |" % (names[k], v["ms"], st[k], alg, pm, v["GBps"], v["TFLOPs"]))
print("| whole forward | %.2f (wall) | %.0f (sum) | %.1f GB | %.1f GB | %.2f TB/s | %.0f |"
      % (r["ms_per_step"], ts, tb / 1e3, tt / 1e3, tb / 1e3 / r["ms_per_step"], r["forward_tflops"]))
print(r["value"], r["ms_per_step"], r["roofline"])
print({k: (v["eager_forward"]["p50"], v["graph_forward_match"]["p50"], v["graph_forward_match_bank5"]["p50"], v["graph_tta_forward_match"]["p50"]) for k, v in r["latency_us"].items()})
print(r["search"]["value"], r["search"]["ms_per_batch"], r["search"]["roofline"]["frac"])
print(r.get("search_1m"))
print(r.get("roofline_by_class"))
print(r["cpu_baseline"])
print(r["other_call_shapes"], r["verification_12k_pairs"])
