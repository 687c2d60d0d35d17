"""ctypes binding of libdiffreg_hip.so (C ABI declared in include/diffreg_hip.h).

The library is the product: there is NO CPU fallback.  Importing this module on a box where the
library was not built raises ImportError; calling an op with CPU tensors raises RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffreg_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError("libdiffreg_hip.so is missing (%s): run `python __graft_entry__.py build` or "
                      "`make -C diff-reg_amd/csrc`" % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH)

c_int, c_size_t, c_void_p, c_char_p, c_double, c_float = (ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p,
                                                          ctypes.c_char_p, ctypes.c_double, ctypes.c_float)

# name -> (restype, argtypes); kept in the order of include/diffreg_hip.h
SIGNATURES = {
    "dr_version": (c_int, []),
    "dr_strerror": (c_char_p, [c_int]),
    "dr_last_hip_error": (c_char_p, []),
    "dr_sinkhorn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dr_sinkhorn_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_sinkhorn_f64": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
}


def _bind(table):
    for name, (res, args) in table.items():
        fn = getattr(_lib, name)
        fn.restype = res
        fn.argtypes = args


_bind(SIGNATURES)

SK_OUT_CONF, SK_OUT_LOG, SK_MINSHIFT, SK_APPLY_MASK, SK_OUT_F32, SK_STRICT = 0x0, 0x1, 0x2, 0x4, 0x8, 0x10


def raw():
    return _lib


def check(code):
    if code != 0:
        msg = _lib.dr_strerror(code).decode()
        if code == -2:
            msg += ": " + _lib.dr_last_hip_error().decode()
        raise RuntimeError("libdiffreg_hip: %s (%d)" % (msg, code))


def ptr(t):
    """device pointer of a contiguous ROCm tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libdiffreg_hip ops need tensors on a ROCm device (got %s); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("libdiffreg_hip ops need contiguous tensors")
    return c_void_p(t.data_ptr())


def stream_of(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def mask_u8(m):
    if m is None:
        return None
    if m.dtype == torch.bool:
        return m.contiguous().view(torch.uint8)
    return (m != 0).contiguous().view(torch.uint8)


def sinkhorn(scores, bin_score, iters, src_mask=None, tgt_mask=None, *, minshift=False, apply_mask=False,
             log_output=False, out_f32=False, strict=False, out=None):
    """Batched Sinkhorn with dustbins on [B,N,M] fp32/fp64 scores.

    Returns conf [B,N,M] (= exp(log_optimal_transport(...))[:, :-1, :-1]) or, with log_output, the full
    [B,N+1,M+1] log assignment (3D/models/matching.py:61-93)."""
    assert scores.dim() == 3
    scores = scores.contiguous()
    B, N, M = scores.shape
    f64 = scores.dtype == torch.float64
    if not f64 and scores.dtype != torch.float32:
        raise RuntimeError("sinkhorn: fp32 or fp64 scores only")
    flags = (SK_OUT_LOG if log_output else 0) | (SK_MINSHIFT if minshift else 0) | \
            (SK_APPLY_MASK if apply_mask else 0) | (SK_OUT_F32 if (f64 and out_f32) else 0) | (SK_STRICT if strict else 0)
    odt = torch.float32 if (not f64 or out_f32) else torch.float64
    shape = (B, N + 1, M + 1) if log_output else (B, N, M)
    if out is None:
        out = torch.empty(shape, dtype=odt, device=scores.device)
    else:
        assert out.shape == shape and out.dtype == odt and out.is_contiguous()
    bs = bin_score.detach().to(device=scores.device, dtype=torch.float32).reshape(1).contiguous()
    wsb = _lib.dr_sinkhorn_workspace_bytes(B, N, M, 8 if f64 else 4, flags)
    ws = torch.empty(wsb, dtype=torch.uint8, device=scores.device) if wsb else None
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    fn = _lib.dr_sinkhorn_f64 if f64 else _lib.dr_sinkhorn_f32
    check(fn(B, N, M, ptr(scores), ptr(sm), ptr(tm), ptr(bs), int(iters), flags, ptr(out), ptr(ws), wsb,
             stream_of(scores)))
    return out
