"""The shared library loads without a GPU and exports every function include/face_recognition_amd.h
declares; argument validation that happens before any GPU call works on the CPU-only host."""
import ctypes as C
import os
import re

from face_recognition_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "face_recognition_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:fra|mfn|fb|ev|fa)_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 24
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of sync with the header"


def test_library_loads_and_reports_version():
    lib = _lib.load()
    assert lib.fra_version() == 1


def test_pure_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.mfn_create(None, 0, 512, _lib.FRA_F16) == 1
    assert b"null" in lib.fra_last_error()
    h = C.c_void_p()
    assert lib.mfn_create(C.byref(h), 0, 500, _lib.FRA_F16) == 1 and b"embedding_size" in lib.fra_last_error()
    assert lib.mfn_create(C.byref(h), 0, 512, 7) == 1
    assert lib.fb_create(C.byref(h), 0, None, 0, 0, 512, _lib.FRA_F16, 0) == 1
    assert lib.fb_create(C.byref(h), 0, None, 0, 10, 100, _lib.FRA_F16, 0) == 1 and b"D ==" in lib.fra_last_error()
    assert lib.fb_create(C.byref(h), 0, None, 0, 10, 512, _lib.FRA_F16, 2 ** 31) == 1
    assert lib.mfn_destroy(None) == 0 and lib.fb_destroy(None) == 0
    assert lib.mfn_op_count(None) == -1 and lib.fb_rows(None) == -1
