"""ctypes binding of libhamspine_hip.so (the C ABI declared in include/hamspine.h).

The product path has no CPU fallback: if the shared library is missing, or a tensor that reaches a
kernel wrapper is not on the HIP device, we raise.  (The CPU oracle lives under /oracle and is only
used by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhamspine_hip.so")

HS_OK = 0
HS_F32, HS_BF16 = 0, 1
A_KC, A_RC, A_CONV, A_DGRAD = 0, 1, 2, 3
B_KC, B_RC, B_WDGRAD, B_CONV = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
MUL_NONE, MUL_GELU_GRAD, MUL_RELU_MASK = 0, 1, 2


class HamspineError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "C", "P", "Q", "K", "R", "S", "stride", "pad",
        "row_pitch", "img_pitch", "qstep", "no_bounds")]


class GemmParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("a_kind", C.c_int32), ("b_kind", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", C.c_void_p), ("B", C.c_void_p),
        ("a_elems", C.c_int64), ("b_elems", C.c_int64),
        ("lda", C.c_int32), ("ldb", C.c_int32),
        ("g", ConvGeom),
        ("batch", C.c_int32), ("batch_inner", C.c_int32),
        ("a_bs0", C.c_int64), ("a_bs1", C.c_int64), ("b_bs0", C.c_int64), ("b_bs1", C.c_int64),
        ("d_bs0", C.c_int64), ("d_bs1", C.c_int64),
        ("split_k", C.c_int32), ("splitk_ws", C.c_void_p),
        ("D", C.c_void_p), ("ldd", C.c_int32), ("out_dtype", C.c_int32),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("act", C.c_int32),
        ("D_preact", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("mul_mode", C.c_int32), ("mul_src", C.c_void_p), ("ldm", C.c_int32),
        ("accumulate", C.c_int32),
    ]


_lib = None


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HamspineError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.hs_last_error.restype = C.c_char_p
        _lib.hs_gemm_splitk_ws_bytes.restype = C.c_int64
        _declare(_lib)
    return _lib


def _declare(l):
    l.hs_gemm.argtypes = [C.POINTER(GemmParams), C.c_void_p]
    l.hs_gemm_splitk_ws_bytes.argtypes = [C.POINTER(GemmParams)]
    l.hs_gemm_suggest_split.argtypes = [C.c_int32] * 4


def check(status, what=""):
    if status != HS_OK:
        msg = lib().hs_last_error().decode("utf-8", "replace")
        raise HamspineError(f"{what} failed (status {status}): {msg}")


def exported_symbols():
    """Names declared in include/hamspine.h (used by the CPU-side ABI test)."""
    import re
    hdr = os.path.join(_HERE, "..", "..", "include", "hamspine.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", txt)))
