"""Timing-only ablations of the fused block kernel (never shipped; their outputs are wrong on purpose).

Writes build_ab/abl_<name>.hip: kernels_block.hip with one part of the work removed.  Build each with
scripts/build_alt.sh, then compare on ONE box:
    FRA_BENCH_UNCHECKED=1 bash scripts/gpu_ab.sh "build_ab/<name>.so ..." bash scripts/gpu_fwd.sh
The time a removal saves is what that part costs where it stands (not its stand-alone cost).
Accumulators are kept alive where a removal would let the compiler drop the matrix work feeding them.
"""
import os

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "face_recognition_amd/csrc/kernels_block.hip")).read()
out = os.path.join(root, "build_ab")
os.makedirs(out, exist_ok=True)


def variant(name, pairs):
    v = src
    for old, new in pairs:
        assert old in v, (name, old)
        v = v.replace(old, new)
    open(os.path.join(out, "abl_%s.hip" % name), "w").write(v)
    print("wrote", name)


KEEP = '        for (int st = 0; st < 2; ++st) for (int n = 0; n < NT16; ++n) asm volatile("" :: "v"(pacc[st][n]));\n'

variant("noproj", [  # no project MFMAs, no project weight reads
    ("                    project<ST, (q >> 1)>(pacc);", '                    asm volatile("" :: "v"(bf[0]), "v"(bf[1]));'),
    ("            if constexpr ((q & 1) && p == 2) issue_a3_first<(q >> 1)>();", ""),
    ("        if ((q & 1) && p == 2)\n            for (int i = 0; i < cmin(4, nt16); ++i) { issued += 1; a3_idx[i] = issued - 1; }", ""),
    ("                    if (i + 4 < nt16) { issued += 1; a3_idx[i + 4] = issued - 1; }", ""),
])
variant("nodwmfma", [  # no depthwise MFMAs (reads and waits stay)
    ("                acc[0] = mfma16(dg, er[slot][0], bias);\n                acc[1] = mfma16(dg, er[slot][1], bias);",
     '                acc[0] = bias; acc[1] = bias; asm volatile("" :: "v"(dg), "v"(er[slot][0]), "v"(er[slot][1]));'),
    ("                acc[0] = mfma16(dg, er[slot][0], acc[0]);\n                acc[1] = mfma16(dg, er[slot][1], acc[1]);",
     '                asm volatile("" :: "v"(dg), "v"(er[slot][0]), "v"(er[slot][1]));'),
])
variant("noprelu", [("                        bf[s][4 * (q & 1) + i] = (T)fmaf(neg_part(d), slope[i], d);   // PReLU: slope holds slope - 1",
                     "                        bf[s][4 * (q & 1) + i] = (T)d;")])
variant("nodwstage", [("            dwproject(c, c & 1, 0);\n", "")])
variant("noexpand", [("            expand(c, c & 1, 0);\n", "")])
variant("nozero", [("        {\n            constexpr int UB = 2 * 8,", "        if (img == (int)blockIdx.y) {\n            constexpr int UB = 2 * 8,")])
variant("noshortcut", [
    ("            if constexpr (RESID && k < EPI_NCH) {", "            if constexpr (false && k < EPI_NCH) {"),
    ("                        if (RESID) {\n    #pragma unroll\n                            for (int i = 0; i < 4; ++i) v[i] += (float)rr[k & 1][j][i];\n                        }", ""),
])
variant("lessstagew", [(  # only buffer 1 (which the epilogue overwrites) is staged again, once per image
    "            if (c + 1 < NCH) stage_w(c + 1, (c + 1) & 1);  // that buffer was last read before the previous barrier\n            else if (has_next) stage_w(0, 0);\n",
    "            if (c == 0) stage_w(1, 1);\n")])
variant("noepilogue", [
    ("        fetch_shortcut(std::integral_constant<int, 0>{});\n        if (dw_active) {", KEEP + "        if (false) {"),
    ("        __syncthreads();\n        {\n            const f32x4 bias = *reinterpret_cast<const f32x4*>(b3 + 4 * g);",
     "        if (B < 0) {\n            const f32x4 bias = *reinterpret_cast<const f32x4*>(b3 + 4 * g);"),
])
variant("nostore", [("                        *reinterpret_cast<V4*>(os + (oo + it * PSTEP * COUT)) = o;",
                     '                        if (B < 0) *reinterpret_cast<V4*>(os + (oo + it * PSTEP * COUT)) = o; else asm volatile("" :: "v"(o));')])
