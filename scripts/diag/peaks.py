"""Sustained copy / read bandwidth and f16 MFMA rate of this GPU (one JSON line); see peaks.hip."""
import ctypes as C, json, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libpeaks.so"))
lib.peaks_copy_ms.restype = C.c_float; lib.peaks_copy_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.peaks_read_ms.restype = C.c_float; lib.peaks_read_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.peaks_mfma_ms.restype = C.c_float; lib.peaks_mfma_ms.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
dev = torch.device("cuda:0")
nbytes = 4 << 30
src = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
dst = torch.empty_like(src)
sink = torch.zeros(16, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
out = {"buffer_GiB": nbytes / 2 ** 30}
ms = lib.peaks_copy_ms(src.data_ptr(), dst.data_ptr(), nbytes, 20)
out["copy_GBps_read_plus_write"] = round(2 * nbytes / (ms * 1e-3) / 1e9, 1)
ms = lib.peaks_read_ms(src.data_ptr(), sink.data_ptr(), nbytes, 20)
out["read_GBps"] = round(nbytes / (ms * 1e-3) / 1e9, 1)
cus = torch.cuda.get_device_properties(0).multi_processor_count
for kind, name in ((0, "mfma_f16_32x32x16_TFLOPs"), (1, "mfma_f16_16x16x32_TFLOPs")):
    for waves_per_simd in (1, 2):
        blocks, iters = cus * waves_per_simd, 20000
        ms = lib.peaks_mfma_ms(kind, sink.data_ptr(), blocks, iters, 5)
        out[f"{name}_{waves_per_simd}wave_per_simd"] = round(blocks * 4 * iters * 131072 / (ms * 1e-3) / 1e12, 1)
out["compute_units"] = cus
print(json.dumps(out))
