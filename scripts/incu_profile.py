#!/usr/bin/env python3
"""In-CU floors per kernel of the forward at batch 1024 -> profiles/<round>_incu.json (read by bench.py).

    python scripts/incu_profile.py r03 [gpurun_out/incu_counters.json]

Inside a compute unit three pipes can bound a fused kernel: the matrix pipe (per SIMD), the LDS array (one per compute
unit, 64 banks x 4 bytes per clock) and the vector ALU (per SIMD).  Per kernel this script combines
  * the EXECUTED instruction counts of one launch (rocprofv3 PMC passes, scripts/gpu_incu.sh: SQ_INSTS_MFMA,
    SQ_INSTS_LDS_LOAD / _STORE, SQ_INSTS_VALU, summed over the chip), and
  * what one instruction of each kind costs, from the DISASSEMBLY of the library's gfx950 code object: the kernel's static
    mix of matrix opcodes (a 16x16x32 or sparse 16x16x64 MFMA holds its SIMD's matrix pipe for 16 clocks, a 32x32x16 for
    32) and of LDS opcodes (array clocks per wave instruction: 16-byte reads 4, 8- and 4-byte reads 2, writes 2 per 4
    bytes of a lane) -- the inner loops are unrolled straight-line code, so the static mix is the executed mix,
into clocks per launch and pipe:  mfma = N_mfma x clocks / SIMDs,  lds = (N_load x clocks + N_store x clocks) / CUs,
valu = 2 x (N_valu - N_mfma) / SIMDs (a wave64 instruction takes the SIMD's 32 lanes two clocks).  The file also keeps the
pipes' BUSY clocks as the counters report them (SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_IDX_ACTIVE less SQ_LDS_BANK_CONFLICT,
SQ_ACTIVE_INST_VALU) so that the model can be checked against them.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "face_recognition_amd", "libface_recognition_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
CUS, SIMDS = 256, 1024

MFMA_CLOCKS = [(re.compile(r"v_mfma_f32_32x32x16"), 32), (re.compile(r"v_mfma_f32_16x16x32"), 16),
               (re.compile(r"v_smfmac_f32_16x16x64"), 16), (re.compile(r"v_mfma|v_smfmac"), 16)]
# LDS array clocks per wave instruction (64 lanes through 64 banks of 4 bytes)
LDS_LOAD = {"ds_read_b128": 4, "ds_read_b96": 8, "ds_read_b64": 2, "ds_read_b32": 2, "ds_read_u16": 2, "ds_read_u8": 2,
            "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_read_u16_d16": 2, "ds_read_u16_d16_hi": 2, "ds_read2st64_b32": 4,
            "ds_read_b64_tr_b16": 2, "ds_read_i8": 2, "ds_read_i16": 2, "ds_bpermute_b32": 2, "ds_read2st64_b64": 8}
LDS_STORE = {"ds_write_b128": 8, "ds_write_b96": 8, "ds_write_b64": 4, "ds_write_b32": 2, "ds_write_b16": 2, "ds_write_b8": 2,
             "ds_write2_b32": 4, "ds_write2_b64": 8, "ds_write_b16_d16_hi": 2, "ds_write2st64_b32": 4, "ds_write2st64_b64": 8}


def disassemble():
    """{mangled kernel name: list of mnemonics} for every gfx950 code object bundled into the library."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)] + [len(blob)]
        for n, (a, b) in enumerate(zip(starts, starts[1:])):
            part, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
            open(part, "wb").write(blob[a:b])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = out.setdefault(m.group(1), [])
                    continue
                m = re.match(r"^\s+(\w+)", line)
                if m and cur is not None:
                    cur.append(m.group(1))
    return out


def static_costs(mnemonics):
    mf = [next(c for rx, c in MFMA_CLOCKS if rx.match(m)) for m in mnemonics if m.startswith(("v_mfma", "v_smfmac"))]
    ld = [LDS_LOAD[m] for m in mnemonics if m in LDS_LOAD]
    st = [LDS_STORE[m] for m in mnemonics if m in LDS_STORE]
    unknown = sorted({m for m in mnemonics if m.startswith("ds_") and m not in LDS_LOAD and m not in LDS_STORE
                      and not m.startswith(("ds_swizzle", "ds_nop", "ds_add", "ds_max", "ds_min", "ds_or", "ds_and", "ds_inc", "ds_dec"))})
    avg = lambda v: sum(v) / len(v) if v else 0.0   # noqa: E731
    return {"static_mfma": len(mf), "clocks_per_mfma": avg(mf), "static_lds_loads": len(ld), "clocks_per_lds_load": avg(ld),
            "static_lds_stores": len(st), "clocks_per_lds_store": avg(st), "unpriced_lds_opcodes": unknown,
            "static_instructions": len(mnemonics)}


def demangle(name):
    """c++filt of this image does not know the _Float16 / __bf16 manglings (DF16_, DF16b): spell them as types it knows
    and put the compiler's names back."""
    if not name.startswith("_Z"):
        return name
    m = name.replace("DF16b", "u6__bf16").replace("DF16_", "Dh")
    out = subprocess.run(["c++filt", m], capture_output=True, text=True).stdout.strip() or name
    return re.sub(r"\bhalf\b", "_Float16", out)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    counters_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "incu_counters.json")
    counters = json.load(open(counters_path))
    code = {demangle(k): v for k, v in disassemble().items()}
    kernels = {}
    for name, c in counters.items():
        name = demangle(name)
        if name not in code:
            print("no disassembly for", name[:80], file=sys.stderr)
            continue
        s = static_costs(code[name])
        n_mfma, n_valu = c.get("SQ_INSTS_MFMA", 0.0), c.get("SQ_INSTS_VALU", 0.0)
        n_ld, n_st = c.get("SQ_INSTS_LDS_LOAD", 0.0), c.get("SQ_INSTS_LDS_STORE", 0.0)
        model = {"mfma_clocks": n_mfma * s["clocks_per_mfma"] / SIMDS,
                 "lds_clocks": (n_ld * s["clocks_per_lds_load"] + n_st * s["clocks_per_lds_store"]) / CUS,
                 "valu_clocks": 2.0 * max(0.0, n_valu - n_mfma) / SIMDS}
        busy = {"mfma_clocks": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / SIMDS,
                "lds_clocks": (c.get("SQ_LDS_IDX_ACTIVE", 0.0) - c.get("SQ_LDS_BANK_CONFLICT", 0.0)) / CUS,
                "lds_clocks_with_bank_conflicts": c.get("SQ_LDS_IDX_ACTIVE", 0.0) / CUS,
                "valu_active_clocks": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / SIMDS,
                "kernel_clocks": c.get("SQ_BUSY_CYCLES", 0.0) / 32}
        kernels[name] = {"executed_per_launch": {"mfma": n_mfma, "lds_load": n_ld, "lds_store": n_st, "valu": n_valu,
                                                 "salu": c.get("SQ_INSTS_SALU", 0.0), "vmem": c.get("SQ_INSTS_VMEM", 0.0)},
                         "from_disassembly": s, "model_clocks_per_launch": {k: round(v) for k, v in model.items()},
                         "counter_busy_clocks_per_launch": {k: round(v) for k, v in busy.items()}}
        ratio = lambda a, b: a / b if b else float("nan")   # noqa: E731
        print("%-64s mfma %7.0f (busy %7.0f, %.2f)  lds %7.0f (busy %7.0f, %.2f)  valu %7.0f  kernel %7.0f" % (
            name[10:74], model["mfma_clocks"], busy["mfma_clocks"], ratio(model["mfma_clocks"], busy["mfma_clocks"]),
            model["lds_clocks"], busy["lds_clocks"], ratio(model["lds_clocks"], busy["lds_clocks"]), model["valu_clocks"],
            busy["kernel_clocks"]))
    out = {"batch": 1024, "compute_units": CUS, "simds": SIMDS,
           "source": "scripts/gpu_incu.sh (rocprofv3 PMC passes) + scripts/incu_profile.py (disassembly of the library)",
           "kernels": kernels}
    path = os.path.join(ROOT, "profiles", f"{tag}_incu.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
