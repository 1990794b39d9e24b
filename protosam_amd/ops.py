"""Thin tensor -> raw-pointer adapters over the C-ABI kernels. PyTorch is used here only for device
memory and the current HIP stream; all arithmetic happens in ``libprotosam_hip.so``."""
import torch

from . import _lib

EPI_F16, EPI_GELU_F16, EPI_F32 = 0, 1, 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (protosam_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def gemm(a, w, bias=None, out=None, epilogue=EPI_F16, resid=None, gamma=None, resid_mod=0, out_seg=0,
         out_seg_stride=0, out_seg_off=0, M=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias). a/w fp16 (K contiguous); out fp16 or fp32 by epilogue."""
    _req(a, torch.float16, "a"); _req(w, torch.float16, "w")
    _req(bias, torch.float32, "bias"); _req(resid, torch.float32, "resid"); _req(gamma, torch.float32, "gamma")
    a2 = a.reshape(-1, a.shape[-1]) if a.dim() != 2 else a
    if M is None:
        M = a2.shape[0]
    K = a2.shape[1]
    N = w.shape[0]
    assert w.shape[1] == K, (w.shape, K)
    odt = torch.float32 if epilogue == EPI_F32 else torch.float16
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    _req(out, odt, "out")
    out2 = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
    ldr = 0
    if resid is not None:
        r2 = resid.reshape(-1, resid.shape[-1]) if resid.dim() != 2 else resid
        ldr = r2.stride(0)
    st = _lib.lib().psam_gemm_f16(_ptr(a2), _ptr(w), _ptr(bias), _ptr(out2), _ptr(resid), _ptr(gamma), M, N, K,
                                 a2.stride(0), w.stride(0), out2.stride(0), ldr, resid_mod, out_seg,
                                 out_seg_stride, out_seg_off, epilogue, _stream())
    _lib.check(st, "psam_gemm_f16")
    return out


def layernorm(x, weight, bias, eps, out=None, out_dtype=torch.float16, out2=None, zero_tail_rows=0, M=None):
    """Row LayerNorm of fp32 x[M,D]. Optionally writes `zero_tail_rows` all-zero rows after row M-1."""
    _req(x, torch.float32, "x"); _req(weight, torch.float32, "weight"); _req(bias, torch.float32, "bias")
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
    if M is None:
        M = x2.shape[0]
    D = x2.shape[1]
    if out is None:
        out = torch.empty((M + zero_tail_rows, D), dtype=out_dtype, device=x.device)
    o2 = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
    st = _lib.lib().psam_layernorm(_ptr(x2), _ptr(weight), _ptr(bias), _ptr(o2), _ptr(out2), M, D, x2.stride(0),
                                  o2.stride(0), float(eps), 0 if out.dtype == torch.float16 else 1,
                                  zero_tail_rows, _stream())
    _lib.check(st, "psam_layernorm")
    return out


def attention(qkv, B, N, H, hd, scale, out=None, mode=0, rel_h=None, rel_w=None, pad_row=None, gh=0, gw=0, ws=0):
    """qkv fp16 [B,N,3,H,hd] (packed as nn.Linear(dim,3*dim) emits it) -> fp16 [B,N,H*hd]."""
    _req(qkv, torch.float16, "qkv"); _req(rel_h, torch.float32, "rel_h"); _req(rel_w, torch.float32, "rel_w")
    _req(pad_row, torch.float16, "pad_row")
    assert qkv.is_contiguous()
    if out is None:
        out = torch.empty((B, N, H * hd), dtype=torch.float16, device=qkv.device)
    st = _lib.lib().psam_attention_f16(_ptr(qkv), _ptr(out), _ptr(rel_h), _ptr(rel_w), _ptr(pad_row), B, N, H, hd,
                                      float(scale), mode, gh, gw, ws, _stream())
    _lib.check(st, "psam_attention_f16")
    return out


def relpos(qkv, Rh, Rw, B, N, H, hd, gw, K, windowed, rel_h=None, rel_w=None):
    _req(qkv, torch.float16, "qkv"); _req(Rh, torch.float32, "Rh"); _req(Rw, torch.float32, "Rw")
    KO = 16 if windowed else 64
    if rel_h is None:
        rel_h = torch.empty((B, H, N, KO), dtype=torch.float32, device=qkv.device)
    if rel_w is None:
        rel_w = torch.empty((B, H, N, KO), dtype=torch.float32, device=qkv.device)
    assert Rh.is_contiguous() and Rw.is_contiguous()
    st = _lib.lib().psam_relpos(_ptr(qkv), _ptr(Rh), _ptr(Rw), _ptr(rel_h), _ptr(rel_w), B, N, H, hd, gw, K,
                               1 if windowed else 0, _stream())
    _lib.check(st, "psam_relpos")
    return rel_h, rel_w
