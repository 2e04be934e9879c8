"""Error behaviour of the C ABI: bad arguments are RETURNED (negative NS_ERR_* code + thread-local
ns_last_error() text), never thrown, never a crash; the Python wrappers turn them into NSError; nothing is launched."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_returned_not_thrown(dev):
    from nspeech_amd import _lib as L
    from nspeech_amd import ops
    lib = L.lib()
    assert lib.ns_version() > 0
    assert lib.ns_device_arch().decode().startswith("gfx")
    # null parameter block
    assert lib.ns_gemm(None, None) < 0 and b"null" in lib.ns_last_error()
    # null operands / bad dtype / split-K without atomics
    p = L.GemmParams()
    p.M = p.N = p.K = 16
    assert lib.ns_gemm(C.byref(p), None) < 0 and b"null operand" in lib.ns_last_error()
    a = torch.zeros(256, device="cuda")
    p.A = p.B = p.C = a.data_ptr()
    p.dtype = 7
    assert lib.ns_gemm(C.byref(p), None) < 0 and b"dtype" in lib.ns_last_error()
    p.dtype, p.c_dtype, p.split_k, p.accumulate = 0, 0, 4, 0
    assert lib.ns_gemm(C.byref(p), None) < 0 and b"split_k" in lib.ns_last_error()
    # wrappers raise NSError with the library's message
    with pytest.raises(L.NSError, match="accumulate needs fp32 C"):
        ops.gemm(a.bfloat16(), a.bfloat16(), a.bfloat16(), 16, 16, 16, 16, 16, 16, accumulate=1)
    # unsupported shapes of the persistent kernels are refused before anything is launched
    z = torch.zeros(64, device="cuda")
    zb = z.bfloat16()
    sp = ops.lstm_seq_params(4, 3, 100, 8, 2, z, 400, zb, None, None, False, zb, 100, z, zb)      # H % 64 != 0
    w = torch.zeros(1 << 16, device="cuda")
    assert not ops.lstm_cluster_supported(sp)
    with pytest.raises(L.NSError, match="H % 64"):
        ops.lstm_cluster("fwd", sp, sp, w)
    torch.cuda.synchronize()                      # nothing faulted
    # a good call still works afterwards and clears nothing it should not
    c = torch.zeros(256, device="cuda")
    ops.gemm(torch.ones(256, device="cuda"), torch.ones(256, device="cuda"), c, 16, 16, 16, 16, 16, 16)
    assert torch.allclose(c, torch.full_like(c, 16.0))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """The product path has no CPU fallback: without libnspeech_hip.so the loader raises instead of degrading."""
    from nspeech_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None, raising=False)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nowhere.so"), raising=False)
    try:
        with pytest.raises(L.NSError):
            L.lib()
    finally:
        monkeypatch.undo()
        L.lib()
