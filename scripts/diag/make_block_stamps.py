"""Diagnostic build of the fused block kernel with in-kernel time stamps (never shipped).

Writes build_ab/kb_stamps.hip: kernels_block.hip with STAMP(i) calls at the phase boundaries of the
slice loop.  One chosen thread (DBG_T) of every workgroup stores the shader clock at each stamp into
a __device__ array that fra_dbg_read() copies out.
"""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "face_recognition_amd/csrc/kernels_block.hip")).read()
pre = r'''
#define DBG_NS 64
__device__ unsigned long long fra_dbg_stamps[65536 * DBG_NS];
#define DBG_WG (blockIdx.y * gridDim.x + blockIdx.x)
#define STAMP(i)                                                                                     \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == DBG_T) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
extern "C" int fra_dbg_read(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fra_dbg_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
'''
assert '#include "block_stages.h"\n' in s
s = s.replace('#include "block_stages.h"\n', '#include "block_stages.h"\n' + pre, 1)
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
# stamps of the first two images a workgroup handles: slots 30*it + i; kernel start in 60, real time in 62/63
rep("    const int band = blockIdx.x;\n",
    "    STAMP(60);\n    if (threadIdx.x == DBG_T) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + 62] = __builtin_amdgcn_s_memrealtime();\n    int dbg_it = 0;\n    const int band = blockIdx.x;\n")
s = s.replace("#define STAMP(i) ", "#define STAMP_RAW(i) ")
s = s.replace("extern \"C\" int fra_dbg_read", "#define STAMP(i) STAMP_RAW(i)\n#define ISTAMP(i) do { if (dbg_it < 2) STAMP_RAW(30 * dbg_it + (i)); } while (0)\nextern \"C\" int fra_dbg_read")
rep("        const bool has_next = img + (int)gridDim.y < B;\n", "        const bool has_next = img + (int)gridDim.y < B;\n        ISTAMP(0);\n")
rep('        if (img == (int)blockIdx.y) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        __syncthreads();\n',
    '        if (img == (int)blockIdx.y) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        ISTAMP(1);\n        __syncthreads();\n        ISTAMP(2);\n')
rep("            if (c == NCH - 1 && has_next) load_x(img + gridDim.y);   // the fragments are dead after the last expand\n",
    "            if (c == NCH - 1 && has_next) load_x(img + gridDim.y);\n            ISTAMP(3 + 5 * c);\n")
rep('            asm volatile("" ::: "memory");\n            dwproject(c, c & 1, 0);\n            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n            __syncthreads();                           // image and old weights free, new weights visible\n',
    '            asm volatile("" ::: "memory");\n            ISTAMP(4 + 5 * c);\n            dwproject(c, c & 1, 0);\n            ISTAMP(5 + 5 * c);\n            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n            ISTAMP(6 + 5 * c);\n            __syncthreads();\n            ISTAMP(7 + 5 * c);\n')
rep("        __syncthreads();\n        {\n            const f32x4 bias =", "        ISTAMP(24);\n        __syncthreads();\n        ISTAMP(25);\n        {\n            const f32x4 bias =")
rep("        if (has_next) __syncthreads();   // staging reads are done before the region is zeroed and restaged\n",
    "        ISTAMP(26);\n        if (has_next) __syncthreads();\n        ISTAMP(27);\n        ++dbg_it;\n")
rep("        if (has_next) __syncthreads();\n        ISTAMP(27);\n        ++dbg_it;\n    }\n}",
    "        if (has_next) __syncthreads();\n        ISTAMP(27);\n        ++dbg_it;\n    }\n    STAMP(61);\n    if (threadIdx.x == DBG_T) fra_dbg_stamps[(size_t)DBG_WG * DBG_NS + 63] = __builtin_amdgcn_s_memrealtime();\n}")
out = os.path.join(root, "build_ab")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "kb_stamps.hip"), "w").write(s)
print("wrote", os.path.join(out, "kb_stamps.hip"))
