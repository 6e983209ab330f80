"""Diagnostic build of the stem kernel with in-kernel time stamps (never shipped): build_ab/ks_stamps.hip."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "face_recognition_amd/csrc/kernels_stem.hip")).read()
pre = r'''
#define DBG_NS 64
__device__ unsigned long long fra_dbg_stamps[65536 * DBG_NS];
#define DBG_WG (blockIdx.x)
#define STAMP(i)                                                                                     \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == DBG_T && DBG_WG < 65536) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#define STAMPI(i) do { if (item == (int)(blockIdx.x + 5 * gridDim.x)) STAMP(i); } while (0)
extern "C" int fra_dbg_read(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fra_dbg_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
'''
s = s.replace('#include "common.h"\n', '#include "common.h"\n' + pre, 1)
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
rep("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n", "    STAMP(60);\n    if (threadIdx.x == DBG_T && DBG_WG < 65536) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + 62] = __builtin_amdgcn_s_memrealtime();\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n")
# stamps inside the item loop are taken at the workgroup's sixth item
rep("        const int y0 = band * TR, row0 = y0 - 1;\n", "        const int y0 = band * TR, row0 = y0 - 1;\n        STAMPI(0);\n")
rep("        __syncthreads();\n        if (item + (int)gridDim.x < n_items) request(", "        __syncthreads();\n        STAMPI(1);\n        if (item + (int)gridDim.x < n_items) request(")
rep("        // ================= stem: im2col GEMM", "        STAMPI(2);\n        // ================= stem: im2col GEMM")
rep("        __syncthreads();\n        if (item + (int)gridDim.x < n_items) stage(", "        STAMPI(3);\n        __syncthreads();\n        STAMPI(4);\n        if (item + (int)gridDim.x < n_items) stage(")
rep("        // ================= depthwise 3x3 + PReLU", "        STAMPI(5);\n        // ================= depthwise 3x3 + PReLU")
rep("            dw.preamble();\n            dw.run();\n        }\n    }\n}", "            dw.preamble();\n            dw.run();\n        }\n        STAMPI(6);\n    }\n    STAMP(61);\n    if (threadIdx.x == DBG_T && DBG_WG < 65536) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + 63] = __builtin_amdgcn_s_memrealtime();\n}")
out = os.path.join(root, "build_ab")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "ks_stamps.hip"), "w").write(s)
print("wrote ks_stamps.hip")
