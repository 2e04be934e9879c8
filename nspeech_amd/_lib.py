"""ctypes binding of libnspeech_hip.so (the C ABI declared in include/nspeech_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
fails, this module raises.  PyTorch only supplies device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnspeech_hip.so")

NS_F32, NS_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3

_lib = None


class NSError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; fail loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NSError(
                "libnspeech_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the hot path)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.ns_last_error.restype = C.c_char_p
        _lib.ns_device_arch.restype = C.c_char_p
        _lib.ns_version.restype = C.c_int
    return _lib


def check(rc, what):
    if rc != 0:
        raise NSError("%s failed (%d): %s" % (what, rc, lib().ns_last_error().decode()))


def call(name, params, stream):
    """Invoke `int ns_<name>(const params*, hipStream_t)`."""
    fn = getattr(lib(), name)
    rc = fn(C.byref(params), C.c_void_p(stream))
    check(rc, name)


class GemmParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_mode", C.c_int),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_mode", C.c_int),
        ("b_seg_len", C.c_int), ("b_seg_stride", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_dtype", C.c_int),
        ("accumulate", C.c_int),
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("alpha", C.c_float),
        ("row_period", C.c_int), ("row_lo", C.c_int), ("row_hi", C.c_int), ("row_shift", C.c_int),
        ("col_sum", C.c_void_p), ("col_sumsq", C.c_void_p),
        ("split_k", C.c_int),
    ]
