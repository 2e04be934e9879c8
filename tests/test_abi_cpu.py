"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every entry point that
include/nspeech_hip.h declares; the Python structs are generated from that same header."""
import ctypes
import os

from nspeech_amd import _lib


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.lib()
    assert len(_lib.FUNCS) >= 25
    missing = [f for f in _lib.FUNCS if not hasattr(lib, f)]
    assert not missing, missing
    assert lib.ns_device_arch() == b"gfx950"
    assert lib.ns_version() >= 100


def test_structs_follow_the_header():
    g = _lib.struct("ns_gemm_params")
    names = [f[0] for f in g._fields_]
    for want in ("dtype", "M", "N", "K", "A", "lda", "a_mode", "B", "b_seg_stride", "col_sumsq", "split_k",
                 "f32_passes"):
        assert want in names
    assert ctypes.sizeof(g) % 8 == 0
    a = _lib.struct("ns_taco2_attn_params")
    assert {"keys_t", "align_t", "dctx_t", "work"} <= {f[0] for f in a._fields_}


def test_bad_arguments_return_errors_not_crashes():
    # argument validation happens on the host before any launch, so it is testable without a GPU
    lib = _lib.lib()
    p = _lib.struct("ns_gemm_params")
    p.M, p.N, p.K = 4, 4, 4          # null operands
    rc = lib.ns_gemm(ctypes.byref(p), None)
    assert rc == -1 and b"null operand" in lib.ns_last_error()
    q = _lib.struct("ns_lstm_seq_params")
    assert lib.ns_lstm_seq_fwd(ctypes.byref(q), None) == -1


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nspeech_amd import hparams
    from nspeech_amd.utils import audio
    hparams.load("taco2")
    with pytest.raises(_lib.NSError):
        audio.spectrogram([0.0] * 4000)
