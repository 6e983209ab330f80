"""Compile the HIP sources under csrc/ into libface_recognition_amd.so next to this file.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
shared object is what the GPU box loads.  ``python -m face_recognition_amd.build`` rebuilds.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libface_recognition_amd.so")
SOURCES = ["api.cpp", "kernels_conv.hip", "kernels_gemm.hip", "kernels_block.hip", "kernels_rblock.hip", "kernels_wblock.hip", "kernels_pblock.hip", "kernels_stem.hip", "mfn_model.hip", "fb_search.hip", "ev_fold.hip", "fa_align.hip", "kernels_tail.hip"]
ARCH = "gfx950"
# per-source flags.  kernels_wblock.hip runs one wave per SIMD with the whole register file: its compiler-issued MFMAs
# keep their results in ordinary registers (the project sums sit in accumulation registers through inline assembly)
EXTRA_FLAGS = {"kernels_wblock.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc and PATH)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "block_stages.h"), os.path.join(HERE, "..", "include", "face_recognition_amd.h")]
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-x", "hip", "-Wall", "-Wno-unused-function",
             # keep scalar f16*f16+f32 FMAs (v_fma_mix_f32) instead of convert + packed-f32 FMA
             "-fno-slp-vectorize"]
    # the library alone travels to the GPU box (objects stay behind): when it is newer than every source, it is current
    if not force and not _stale(LIB, [os.path.join(CSRC, src) for src in SOURCES] + headers):
        return LIB
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
