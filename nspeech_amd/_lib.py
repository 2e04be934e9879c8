"""ctypes binding of libnspeech_hip.so (the C ABI declared in include/nspeech_hip.h).

The parameter structs are generated from the header itself at import time, so the Python
side cannot drift from the C side.  The product path has NO CPU fallback: if the shared
library is missing or a call fails, this module raises.  PyTorch only supplies device
memory and streams.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnspeech_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nspeech_hip.h")

NS_F32, NS_BF16 = 0, 1
NS_ERR_SHORT_BUFFER = -4
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_SOFTSIGN = 0, 1, 2, 3, 4

_lib = None


class NSError(RuntimeError):
    pass


_CTYPES = {
    "int": C.c_int, "float": C.c_float, "double": C.c_double, "int64_t": C.c_int64,
    "int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "size_t": C.c_size_t,
}


def _parse_header():
    """Return ({struct name: ctypes.Structure subclass}, [exported function names])."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.+)$", decl)
            base, star, names = m.group(2), m.group(3), m.group(4)
            for nm in names.split(","):
                nm = nm.strip()
                is_ptr = bool(star) or nm.startswith("*")
                nm = nm.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)
                if is_ptr:
                    ct = C.c_void_p
                elif base in structs:           # a parameter block embedded in another one
                    ct = structs[base]
                else:
                    ct = _CTYPES[base]
                if arr:
                    nm = arr.group(1)
                    ct = ct * int(arr.group(2))
                fields.append((nm, ct))
        structs[name] = type(name, (C.Structure,), {"_fields_": fields})
    funcs = re.findall(r"\b(ns_\w+)\s*\(", src)
    funcs = sorted(set(f for f in funcs if not f.endswith("_params")))
    return structs, funcs


STRUCTS, FUNCS = _parse_header()


def struct(name):
    return STRUCTS[name]()


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


def GemmParams():
    return struct("ns_gemm_params")
