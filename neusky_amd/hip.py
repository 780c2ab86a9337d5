"""ctypes binding of libneusky_hip.so (the C ABI declared in include/neusky_hip.h).

Fails loudly when the shared library is missing: the product path has no fallback.  Tensors are
passed as raw device pointers (`tensor.data_ptr()`), the stream as torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libneusky_hip.so")


class NeuSkyHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
        "neusky_amd has no CPU/PyTorch fallback."
    )
_lib = C.CDLL(LIB_PATH)

# ---- epilogues (keep in sync with include/neusky_hip.h)
EPI_NONE, EPI_RELU, EPI_LEAKY, EPI_SIGMOID, EPI_SOFTPLUS, EPI_FILM, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
EPI_BWD_RELU, EPI_BWD_LEAKY, EPI_BWD_FILM, EPI_EXP = 7, 8, 9, 10


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("a_kcontig", C.c_int32), ("b_kcontig", C.c_int32),
        ("bias", C.c_void_p),
        ("epi", C.c_int32), ("p0", C.c_float), ("p1", C.c_float),
        ("aux0", C.c_void_p), ("ldaux0", C.c_int32),
        ("aux1", C.c_void_p), ("ldaux1", C.c_int32),
        ("aux2", C.c_void_p), ("ldaux2", C.c_int32),
        ("out1", C.c_void_p), ("ldout1", C.c_int32),
        ("out2", C.c_void_p), ("ldout2", C.c_int32),
        ("row_mod", C.c_int32), ("k_splits", C.c_int32), ("beta", C.c_float),
    ]


_lib.nsky_last_error.restype = C.c_char_p
_lib.nsky_abi_version.restype = C.c_int


def _sig(name, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = C.c_int
    fn.argtypes = list(argtypes)
    return fn


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NeuSkyHipError(f"{what} failed ({rc}): {_lib.nsky_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8), (t.device, t.dtype)
    return t.data_ptr()


def ld(t):
    """leading dimension (elements) of a 2-D row-major view whose rows are contiguous"""
    if t is None:
        return 0
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0)


_gemm = _sig("nsky_gemm_f32", C.POINTER(GemmDesc), C.c_void_p)
_colsum = _sig("nsky_colsum_f32", C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)


def gemm(A, B, Cout, M, N, K, *, a_kcontig=True, b_kcontig=True, bias=None, epi=EPI_NONE, p0=0.0, p1=0.0,
         aux0=None, aux1=None, aux2=None, out1=None, out2=None, row_mod=0, k_splits=0, beta=0.0):
    """C[M,N] = epi(sum_k A(m,k) B(n,k) + bias).  A/B/C are 2-D row-major views (rows contiguous)."""
    d = GemmDesc(
        A=ptr(A), B=ptr(B), C=ptr(Cout), M=M, N=N, K=K, lda=ld(A), ldb=ld(B), ldc=ld(Cout),
        a_kcontig=int(a_kcontig), b_kcontig=int(b_kcontig), bias=ptr(bias), epi=epi, p0=p0, p1=p1,
        aux0=ptr(aux0), ldaux0=ld(aux0), aux1=ptr(aux1), ldaux1=ld(aux1), aux2=ptr(aux2), ldaux2=ld(aux2),
        out1=ptr(out1), ldout1=ld(out1), out2=ptr(out2), ldout2=ld(out2), row_mod=row_mod, k_splits=k_splits, beta=beta,
    )
    check(_gemm(C.byref(d), stream_ptr()), "nsky_gemm_f32")
    return Cout


def colsum(X, M, N, out):
    check(_colsum(ptr(X), M, N, ld(X), ptr(out), stream_ptr()), "nsky_colsum_f32")
    return out


def abi_version() -> int:
    return _lib.nsky_abi_version()
