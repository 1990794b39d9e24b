"""Thin tensor -> raw-pointer adapters over the C-ABI kernels. PyTorch is used here only for device
memory and the current HIP stream; all arithmetic happens in ``libprotosam_hip.so``."""
import torch

from . import _lib

EPI_F16, EPI_GELU_F16, EPI_F32, EPI_RELU_F16 = 0, 1, 2, 3


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current HIP stream's handle. (torch.cuda.current_stream() builds a Stream object through five Python layers - 3 us a call,
    0.7 ms per slice of a one-slice forward with its ~230 launches; the raw accessors are two C calls)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event bracket around every launch of one C-ABI entry (used by bench.py for the roofline object:
    events are recorded on the stream the kernel is launched on). `work` is the algorithmic FLOPs or bytes per launch."""

    def __init__(self):
        self.records = []  # (start_event, end_event, work)
        self.tags = {}     # record index -> tag

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, work, tag=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((e0, e1, work))
        if tag is not None:
            self.tags[len(self.records) - 1] = tag

    def by_tag(self):
        """{tag: (launches, seconds, work)} over the tagged records (the GEMM timer tags a launch with (M, N, K, epilogue, folded))."""
        out = {}
        for i, tag in self.tags.items():
            a, b, w = self.records[i]
            n, t, f = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, t + a.elapsed_time(b) * 1e-3, f + float(w[0] if isinstance(w, tuple) else w))
        return out

    def summary(self):
        """(n_launches, total_seconds, total_work) -- call after a device synchronize. `work` may be a tuple (bytes, flops):
        the first member is summed here, `summary2` sums the second."""
        t = sum(a.elapsed_time(b) for a, b, _ in self.records) * 1e-3
        return len(self.records), t, float(sum((w[0] if isinstance(w, tuple) else w) for _, _, w in self.records))

    def summary2(self):
        return float(sum(w[1] for _, _, w in self.records if isinstance(w, tuple)))


GEMM_TIMER = None  # set to a KernelTimer to time psam_gemm_f16 launches
# further optional per-entry timers for bench.py's HBM-side roofline entries: name -> KernelTimer, work = ALGORITHMIC bytes
# of the launch (inputs read once + outputs written once). Names: layernorm, alp_sim, prob_argmax, ccl, attention_window,
# attention_global.
TIMERS = {}


def _tstart(name):
    t = TIMERS.get(name)
    return None if t is None else (t, t.start())


def _tstop(h, work):
    if h is not None:
        h[0].stop(h[1], work)

# packed qkv layout between the projection GEMM and the attention kernels: the reference's token-major [B,N,3,H,hd]
# (default) or head-major [3,H,B*N,hd] (PSAM_QKV_HEAD_MAJOR=1: contiguous per-head rows for the attention kernels, but
# scattered 160-byte stores for the GEMM; measured on MI355X at 16 slices: 108.8 / 109.2 vs 109.8 / 109.2 slices/s - a wash)
import os as _os
QKV_HEAD_MAJOR = _os.environ.get("PSAM_QKV_HEAD_MAJOR", "0") != "0"
# window layers: compute the decomposed rel-pos query terms inside the attention kernel (default) instead of a psam_relpos
# launch that writes them to HBM (PSAM_FUSE_WINDOW_RELPOS=0, kept for A/B)
FUSE_WINDOW_RELPOS = _os.environ.get("PSAM_FUSE_WINDOW_RELPOS", "1") != "0"
# global layers: the same inside the assembly global-attention kernel (round 5; attention_fused_relpos)
FUSE_GLOBAL_RELPOS = _os.environ.get("PSAM_FUSE_GLOBAL_RELPOS", "1") != "0"


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
         out_seg_stride=0, out_seg_off=0, M=None, out16=None, stats=None, ln_mr=None, ln_s=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias). a/w fp16 (K contiguous); out fp16 or fp32 by epilogue.
    Folded LayerNorm (psam_gemm_f16_ln): with EPI_F32, `out16` (fp16 [rows, N]) receives half(out) and `stats` (fp32
    [rows, N/64, 2]) per-row partial (sum, sum of squares); with EPI_F16 / EPI_GELU_F16, `ln_mr` (the 6-floats-per-row buffer of
    `ln_finalize`: fp32 (mean, rstd) [rows, 2], then the fp16 MFMA fragments of -mean [rows, 8]) and `ln_s` (the s_ext of
    `fold_layernorm`: fp32 [N], then fp16 fragments [N, 8]) apply out = act(rstd * (acc - mean * ln_s) + bias)."""
    _req(a, torch.float16, "a"); _req(w, torch.float16, "w")
    _req(bias, torch.float32, "bias"); _req(gamma, torch.float32, "gamma")
    _req(resid, torch.float16 if epilogue == EPI_RELU_F16 else torch.float32, "resid")   # epilogue 3 adds a half map
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
    fold = out16 is not None or stats is not None or ln_mr is not None
    if fold:
        _req(out16, torch.float16, "out16"); _req(stats, torch.float32, "stats")
        _req(ln_mr, torch.float32, "ln_mr"); _req(ln_s, torch.float32, "ln_s")
        # extended layouts of the assembly kernels (no size travels across the C ABI): ln_mr = ln_mr_buffer() as psam_ln_finalize
        # fills it (6 floats per row), ln_s = the s_ext of fold_layernorm (N floats + N fp16 fragments of 8 = 5 N floats), stats
        # [rows, N / 64, 2]
        if ln_mr is not None:
            assert ln_s is not None and ln_mr.numel() >= 6 * M and ln_s.numel() >= 5 * N, (ln_mr.numel(), 6 * M, ln_s.numel(), 5 * N)
        if stats is not None:
            assert stats.numel() >= M * (N // 64) * 2 and N % 64 == 0
        if out16 is not None:
            assert out16.shape[-1] >= N and out16.numel() >= M * N
    if epilogue == EPI_F32 and a.device.index not in _GEMM_WS:
        _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    if fold:
        st = _lib.lib().psam_gemm_f16_ln(_ptr(a2), _ptr(w), _ptr(bias), _ptr(out2), _ptr(resid), _ptr(gamma), M, N, K,
                                        a2.stride(0), w.stride(0), out2.stride(0), ldr, resid_mod, out_seg,
                                        out_seg_stride, out_seg_off, epilogue, _ptr(out16),
                                        0 if out16 is None else out16.stride(-2), _ptr(stats), _ptr(ln_mr), _ptr(ln_s),
                                        _stream())
    else:
        st = _lib.lib().psam_gemm_f16(_ptr(a2), _ptr(w), _ptr(bias), _ptr(out2), _ptr(resid), _ptr(gamma), M, N, K,
                                     a2.stride(0), w.stride(0), out2.stride(0), ldr, resid_mod, out_seg,
                                     out_seg_stride, out_seg_off, epilogue, _stream())
    if t0 is not None:
        GEMM_TIMER.stop(t0, 2.0 * M * N * K, tag=(M, N, K, epilogue, 1 if fold else 0))
    _lib.check(st, "psam_gemm_f16_ln" if fold else "psam_gemm_f16")
    return out


_SPLITK = {}


def gemm_splitk_ranges(M, N, K):
    """K ranges `gemm_splitk_ln` would use for x[M,N] += a[M,K] w[N,K]^T on the current device, 0 when the shape does not pay (few 256-tiles
    and a long K are needed: one SAM ViT-H / ViT-L image through mlp.lin2). PSAM_GEMM_SPLITK_ASM=0 switches the path off (A/B)."""
    if _os.environ.get("PSAM_GEMM_SPLITK_ASM", "1") == "0":
        return 0
    key = (M, N, K, torch.cuda.current_device(), DISPATCH_EPOCH)
    if key not in _SPLITK:
        _SPLITK[key] = int(_lib.lib().psam_gemm_splitk_ranges(M, N, K))
    return _SPLITK[key]


def splitk_rows(M):
    """rows of one plane of gemm_splitk_ln's workspace: M rounded up to whole 256-row tiles"""
    return (M + 255) // 256 * 256


def gemm_splitk_ln(a, w, bias, x, ks, ws, ln_w=None, ln_b=None, eps=1e-6, out16=None):
    """x[M,N] (fp32, in place) += a[M,K] @ w[N,K]^T + bias as `ks` K ranges per 256-tile in one launch of the assembly kernel, then one
    pass that sums the ranges (fixed order) and writes out16 = LayerNorm(x) (or fp16(x) when ln_w is None). ws: fp32 scratch of at least
    ks * splitk_rows(M) * N elements owned by the caller (psam_gemm_f16_splitk_ln)."""
    _req(a, torch.float16, "a"); _req(w, torch.float16, "w"); _req(bias, torch.float32, "bias"); _req(x, torch.float32, "x")
    _req(ws, torch.float32, "ws"); _req(ln_w, torch.float32, "ln_w"); _req(ln_b, torch.float32, "ln_b"); _req(out16, torch.float16, "out16")
    M, K = a.shape
    N = w.shape[0]
    assert x.shape == (M, N) and w.shape[1] == K and ws.is_contiguous() and ws.numel() >= ks * splitk_rows(M) * N
    assert out16 is None or (out16.shape[0] >= M and out16.shape[-1] >= N)
    t0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    st = _lib.lib().psam_gemm_f16_splitk_ln(_ptr(a), _ptr(w), _ptr(bias), _ptr(x), M, N, K, a.stride(0), w.stride(0), x.stride(0), ks,
                                           _ptr(ws), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(out16),
                                           0 if out16 is None else out16.stride(-2), _stream())
    if t0 is not None:
        GEMM_TIMER.stop(t0, 2.0 * M * N * K, tag=(M, N, K, EPI_F32, 0))
    _lib.check(st, "psam_gemm_f16_splitk_ln")
    return x


_CU_COUNT = {}


def fold_pays(M, D, device, min_fill=0.8):
    """Whether the folded LayerNorm is worth it for residual-stream GEMMs of M rows x D columns: its producer / consumer epilogues
    live in the 256x256-tile assembly kernels (and, slower, in the HIP kernels), so the launches must fill the CUs with 256-tiles -
    one slice at a time (80 tiles for 256 CUs) the half-tile kernels with separate LayerNorm passes are faster (85 vs 73 slices/s)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CU_COUNT:
        _CU_COUNT[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return ((M + 255) // 256) * (D // 256) >= min_fill * _CU_COUNT[idx]


def ln_mr_buffer(M, device):
    """(mean, rstd) buffer of psam_ln_finalize: fp32 [M][2] followed by the fp16 MFMA fragments of -mean [M][8] (6 floats per row)."""
    return torch.empty(6 * M, dtype=torch.float32, device=device)


def ln_finalize(stats, M, D, eps, mr=None):
    """stats fp32 [rows, D/64, 2] (partial sums from a folded-LayerNorm producer GEMM) -> mr (see ln_mr_buffer): fp32 [rows, 2] =
    (mean, rstd), then fp16 [rows, 8] = {hi, hi, lo, 0 x 5} of -mean (the assembly GEMM's rank-1 correction operand)."""
    _req(stats, torch.float32, "stats")
    if mr is None:
        mr = ln_mr_buffer(M, stats.device)
    _req(mr, torch.float32, "mr")
    assert mr.numel() >= 6 * M and mr.is_contiguous()
    st = _lib.lib().psam_ln_finalize(_ptr(stats), M, D, float(eps), _ptr(mr), _stream())
    _lib.check(st, "psam_ln_finalize")
    return mr


def fold_layernorm(weight, bias, ln_weight, ln_bias):
    """One-time weight transform for a Linear that consumes LayerNorm(x): -> (W' fp16 = half(W * ln_weight), ln_s fp32 [N] =
    row sums of the ROUNDED W', bias' fp32 = bias + W . ln_bias), so that  Linear(LN(x)) = rstd * (x W'^T - mean * ln_s) + bias'."""
    W = weight.detach().float()
    Wp = (W * ln_weight.detach().float()[None, :]).half().contiguous()
    s = Wp.float().sum(1).contiguous()
    b = bias.detach().float() if bias is not None else torch.zeros(W.shape[0], dtype=torch.float32, device=W.device)
    # ln_s as psam_gemm_f16_ln takes it: fp32 [N], then the fp16 MFMA fragments {hi, lo, hi, 0 x 5} of s [N][8] (the assembly
    # GEMM subtracts mean_row * s_col with one rank-1 MFMA per block: hi * hi + lo * hi + hi * lo)
    hi = s.half()
    frag = torch.zeros((s.numel(), 8), dtype=torch.float16, device=s.device)
    frag[:, 0], frag[:, 1], frag[:, 2] = hi, (s - hi.float()).half(), hi
    s_ext = torch.cat([s, frag.reshape(-1).view(torch.float32)]).contiguous()
    return Wp, s_ext, (b + W @ ln_bias.detach().float()).contiguous()


def gemm_heads(a, w, bias, hd, out=None, M=None):
    """The packed qkv projection written head-major: half [N/hd, M, hd] (plane = which*H + h). a/w fp16, K contiguous."""
    _req(a, torch.float16, "a"); _req(w, torch.float16, "w"); _req(bias, torch.float32, "bias")
    a2 = a.reshape(-1, a.shape[-1]) if a.dim() != 2 else a
    if M is None:
        M = a2.shape[0]
    K, N = a2.shape[1], w.shape[0]
    assert w.shape[1] == K and N % hd == 0
    if out is None:
        out = torch.empty((N // hd, M, hd), dtype=torch.float16, device=a.device)
    _req(out, torch.float16, "out")
    assert out.is_contiguous() and out.numel() == M * N
    t0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    st = _lib.lib().psam_gemm_f16_heads(_ptr(a2), _ptr(w), _ptr(bias), _ptr(out), M, N, K, a2.stride(0), w.stride(0), hd,
                                       _stream())
    if t0 is not None:
        GEMM_TIMER.stop(t0, 2.0 * M * N * K)
    _lib.check(st, "psam_gemm_f16_heads")
    return out


_GEMM_WS = {}     # device index -> scratch buffer registered with the library for that device
GEMM_WORKSPACE_MB = int(_os.environ.get("PSAM_GEMM_WORKSPACE_MB", "64"))


def _ensure_gemm_workspace(device):
    """Scratch for the split-K form of the fp32-residual GEMM (csrc/gemm.hip launch8kp_splitk): partial sums of one slice through
    fc2 are 63 MB. Allocated once per DEVICE from torch's allocator and registered with the library for that device (the library
    orders users on different streams through an event); 0 MB = no split-K."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _GEMM_WS:
        n = GEMM_WORKSPACE_MB << 20
        with torch.cuda.device(idx):
            buf = torch.empty(max(n, 16), dtype=torch.uint8, device=torch.device("cuda", idx))
            _lib.check(_lib.lib().psam_gemm_set_workspace(buf.data_ptr() if n else 0, n), "psam_gemm_set_workspace")
        _GEMM_WS[idx] = buf
        _GEMM_WS[None] = buf


DISPATCH_EPOCH = 0     # bumped by every set_* switch below: part of the key of captured graphs (a graph bakes the dispatch in)


def _bump_dispatch():
    global DISPATCH_EPOCH
    DISPATCH_EPOCH += 1


def dispatch_key():
    """what a captured graph of a forward depends on besides shapes and weights: every run-time dispatch switch of this module"""
    return (DISPATCH_EPOCH, QKV_HEAD_MAJOR, FUSE_WINDOW_RELPOS, FUSE_GLOBAL_RELPOS)


def gemm_set_tile(tile):
    """0 auto, 1 = 128x128 (HIP), 11 = 256x256 persistent (HIP), 15 / 16 / 17 = the assembly kernels (see csrc/gemm.hip pick_tile)."""
    _bump_dispatch()
    _lib.check(_lib.lib().psam_gemm_set_tile(int(tile)), "psam_gemm_set_tile")


def gemm_set_option(name, value):
    """Dispatch switches of the GEMM: "asm", "half_tiles", "splitk", "nsplit" (0 / 1), "max_wgs" (cap on the persistent grids,
    0 = none) (see include/protosam_hip.h)."""
    _bump_dispatch()
    _lib.check(_lib.lib().psam_gemm_set_option(name.encode(), int(value)), "psam_gemm_set_option")


def gemm_asm_variant(v):
    """Experiment kernels of the assembly GEMM (library built with GENFLAGS=--experiments); 0 = shipped schedule."""
    _bump_dispatch()
    _lib.check(_lib.lib().psam_gemm_asm_variant(int(v)), "psam_gemm_asm_variant")


def im2col(x, B, H, W, C, kh, kw, stride, dil, pad, ldo=None, out=None):
    """token-major half map [B, H*W, C] -> half [B*Ho*Wo, ldo] (column (ky*kw+kx)*C + c; zero K padding up to ldo)."""
    _req(x, torch.float16, "x")
    assert x.is_contiguous()
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if ldo is None:
        ldo = kh * kw * C
    if out is None:
        out = torch.empty((B * Ho * Wo, ldo), dtype=torch.float16, device=x.device)
    st = _lib.lib().psam_im2col(_ptr(x), B, H, W, C, kh, kw, stride, dil, pad, ldo, _ptr(out), _stream())
    _lib.check(st, "psam_im2col")
    return out, Ho, Wo


def im2col_stem(img, ldo=192, out=None):
    """fp32 NCHW [B,3,H,W] -> half [B*Ho*Wo, ldo] patches of the 7x7 / stride 2 / pad 3 stem conv."""
    _req(img, torch.float32, "img")
    assert img.is_contiguous() and img.shape[1] == 3
    B, _, H, W = img.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B * Ho * Wo, ldo), dtype=torch.float16, device=img.device)
    st = _lib.lib().psam_im2col_stem(_ptr(img), B, H, W, ldo, _ptr(out), _stream())
    _lib.check(st, "psam_im2col_stem")
    return out, Ho, Wo


def maxpool3x3s2(x, B, H, W, C, out=None):
    _req(x, torch.float16, "x")
    assert x.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B * Ho * Wo, C), dtype=torch.float16, device=x.device)
    st = _lib.lib().psam_maxpool3x3s2(_ptr(x), B, H, W, C, _ptr(out), _stream())
    _lib.check(st, "psam_maxpool3x3s2")
    return out, Ho, Wo


def rotate_nearest(img, xg, yg, rt, crop_y, crop_x, out_h, out_w):
    """torchvision-style affine NEAREST resampling of fp32 [B,C,H,W] (protosam_amd/rotate.py); rt = host fp32 [3,2]."""
    _req(img, torch.float32, "img"); _req(xg, torch.float32, "xg"); _req(yg, torch.float32, "yg")
    assert img.is_contiguous() and rt.dtype == torch.float32 and rt.device.type == "cpu" and rt.is_contiguous()
    B, C, H, W = img.shape
    out = torch.empty((B, C, out_h, out_w), dtype=torch.float32, device=img.device)
    st = _lib.lib().psam_rotate_nearest(_ptr(img), _ptr(out), _ptr(xg), _ptr(yg), rt.data_ptr(), B * C, H, W, crop_y, crop_x,
                                        out_h, out_w, _stream())
    _lib.check(st, "psam_rotate_nearest")
    return out


def resize_aa(img, oh, ow):
    """anti-aliased bilinear resize of fp32 [B,C,H,W] (aten _upsample_bilinear2d_aa semantics)."""
    _req(img, torch.float32, "img")
    assert img.is_contiguous()
    B, C, H, W = img.shape
    tmp = torch.empty((B * C, H, ow), dtype=torch.float32, device=img.device)
    out = torch.empty((B, C, oh, ow), dtype=torch.float32, device=img.device)
    st = _lib.lib().psam_resize_aa(_ptr(img), _ptr(tmp), _ptr(out), B * C, H, W, oh, ow, _stream())
    _lib.check(st, "psam_resize_aa")
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
    h = _tstart("layernorm")
    st = _lib.lib().psam_layernorm(_ptr(x2), _ptr(weight), _ptr(bias), _ptr(o2), _ptr(out2), M, D, x2.stride(0),
                                  o2.stride(0), float(eps), 0 if out.dtype == torch.float16 else 1,
                                  zero_tail_rows, _stream())
    _tstop(h, M * D * (4 + out.element_size()) + 8 * D)
    _lib.check(st, "psam_layernorm")
    return out


def attention(qkv, B, N, H, hd, scale, out=None, mode=0, rel_h=None, rel_w=None, relq=None, pad_row=None, gh=0, gw=0,
              ws=0, head_major=False, rpack=None):
    """qkv fp16 [B,N,3,H,hd] (packed as nn.Linear(dim,3*dim) emits it) or, head_major, [3,H,B*N,hd] (gemm_heads)
    -> fp16 [B,N,H*hd]."""
    _req(qkv, torch.float16, "qkv"); _req(rel_h, torch.float32, "rel_h"); _req(rel_w, torch.float32, "rel_w")
    _req(pad_row, torch.float16, "pad_row"); _req(relq, torch.float16, "relq"); _req(rpack, torch.float16, "rpack")
    assert qkv.is_contiguous()
    if out is None:
        out = torch.empty((B, N, H * hd), dtype=torch.float16, device=qkv.device)
    h = _tstart("attention_window" if mode == 2 else "attention_global")
    st = _lib.lib().psam_attention_f16(_ptr(qkv), _ptr(out), _ptr(rel_h), _ptr(rel_w), _ptr(relq), _ptr(rpack), _ptr(pad_row), B, N,
                                      H, hd, float(scale), mode, gh, gw, ws, 1 if head_major else 0, _stream())
    # qkv read once, out written once (fp16) + the global kernel's rel-pos terms (fp32 [B,H,N,64] x 2); work2 = FLOPs
    # 4*B*H*N*Nk*hd with Nk = the window's 196 keys (14x14) or all N keys
    nk = ws * ws if mode == 2 else N
    _tstop(h, (B * N * 4 * H * hd * 2 + (2 * B * H * N * 64 * 4 if (mode == 1 and rel_h is not None) else 0),
               4.0 * B * H * N * nk * hd))
    _lib.check(st, "psam_attention_f16")
    return out


_FUSED_RELPOS = {}


def attention_fused_relpos(B, N, H, hd, gh, gw):
    """True when attention(mode=1) takes `rpack` (pack_rel_tables, global form) in place of rel_h / rel_w: the assembly kernels
    psam_gattn_asm_*_fused compute the decomposed rel-pos terms themselves - no psam_relpos launch, no fp32 [B,H,N,64] x 2 round trip
    (PSAM_FUSE_GLOBAL_RELPOS=0: the two-kernel path, for A/B)."""
    if not FUSE_GLOBAL_RELPOS or QKV_HEAD_MAJOR:
        return False
    key = (B, N, H, hd, gh, gw, DISPATCH_EPOCH)
    if key not in _FUSED_RELPOS:
        _FUSED_RELPOS[key] = bool(_lib.lib().psam_attention_fused_relpos(B, N, H, hd, gh, gw))
    return _FUSED_RELPOS[key]


def attention_set_variant(v):
    """bit 0: V2 softmax in the global kernels (0 = round-1 serial form); bits 1-2: window kernel of the fused rel-pos path
    (0 attn_kernel, 1 wattn_kernel, 2 the persistent wattn_p_kernel); bit 3: the register-staged HIP global kernel; bit 4: the
    DMA-fed HIP global kernel everywhere (neither: the assembly global kernel where it applies, see include/protosam_hip.h).
    Default 5; for A/B and the equivalence tests."""
    _bump_dispatch()
    _lib.check(_lib.lib().psam_attention_set_variant(int(v)), "psam_attention_set_variant")


def pack_rel_tables(rel_pos_h, rel_pos_w, windowed, hd):
    """fp32 (2K-1, hd) tables -> fp16 [2 (h,w)][2 (hi,lo)][RP][HDP] for psam_relpos (one-time weight packing)."""
    RP = 32 if windowed else 128
    HDP = (hd + 31) // 32 * 32
    out = torch.zeros((2, 2, RP, HDP), dtype=torch.float16, device=rel_pos_h.device)
    for t, R in enumerate((rel_pos_h, rel_pos_w)):
        R = R.detach().float()
        assert R.shape[0] <= RP and R.shape[1] == hd
        hi = R.half()
        lo = (R - hi.float()).half()
        out[t, 0, :R.shape[0], :hd] = hi
        out[t, 1, :R.shape[0], :hd] = lo
    if not windowed:
        return out.contiguous()
    # window form: the tables, then the constants of the 14 x 14 window geometry the window-attention kernel reads (csrc/attention.hip
    # wattn_kernel): the one-hot key-side fragments of the folded bias, [13 key tiles][64 lanes][8 k-slots], and a zero row
    return torch.cat([out.reshape(-1), _window_onehot_table().to(out.device), out.new_zeros(128)]).contiguous()


_ONEHOT = None


def _window_onehot_table(ws=14):
    """fp16 [13*64*8]: lane (li, g) of key tile T holds k-slots 8g..8g+7 of its key's row: 1 at slot kh and at slot 14 + kw, where
    key = 32 (T >> 1) + 8 (li >> 2) + 4 (T & 1) + (li & 3) (the MFMA row -> key map of the kernels), zero rows for keys >= 196."""
    global _ONEHOT
    if _ONEHOT is None:
        T = torch.arange(13).view(13, 1, 1)
        lane = torch.arange(64).view(1, 64, 1)
        e = torch.arange(8).view(1, 1, 8)
        li, g = lane & 15, lane >> 4
        key = (T >> 1) * 32 + (li >> 2) * 8 + (T & 1) * 4 + (li & 3)
        slot = g * 8 + e
        hot = ((slot == key // ws) | (slot == ws + key % ws)) & (key < ws * ws)
        _ONEHOT = hot.to(torch.float16).reshape(-1)
    return _ONEHOT


def relpos(qkv, rpack, B, N, H, hd, gw, K, windowed, scale, rel_h=None, rel_w=None, relq=None, head_major=False):
    """global: returns (rel_h, rel_w) fp32 [B,H,N,64]; windowed: returns relq fp16 [B,H,N,2,32] (zero-initialised once)."""
    _req(qkv, torch.float16, "qkv"); _req(rpack, torch.float16, "rpack")
    assert rpack.is_contiguous()
    if windowed:
        if relq is None:
            relq = torch.zeros((B, H, N, 2, 32), dtype=torch.float16, device=qkv.device)
    else:
        if rel_h is None:
            rel_h = torch.empty((B, H, N, 64), dtype=torch.float32, device=qkv.device)
        if rel_w is None:
            rel_w = torch.empty((B, H, N, 64), dtype=torch.float32, device=qkv.device)
    st = _lib.lib().psam_relpos(_ptr(qkv), _ptr(rpack), _ptr(rel_h), _ptr(rel_w), _ptr(relq), B, N, H, hd, gw, K,
                               1 if windowed else 0, float(scale), 1 if head_major else 0, _stream())
    _lib.check(st, "psam_relpos")
    return relq if windowed else (rel_h, rel_w)


# ---- ALP -----------------------------------------------------------------------------------------------
META_NBG, META_NFG, META_FGMODE, META_NCELL_FG = 0, 1, 2, 3


class AlpBank:
    """Fixed-capacity prototype bank + device-side counts (see csrc/alp.hip)."""

    def __init__(self, ncell, C, device):
        self.cap = ncell + 1
        self.C = C
        self.bank = torch.zeros((2 * self.cap, C), dtype=torch.float32, device=device)
        self.meta = torch.zeros(8, dtype=torch.int32, device=device)
        self.slot_bg = torch.empty(ncell, dtype=torch.int32, device=device)
        self.slot_fg = torch.empty(ncell, dtype=torch.int32, device=device)
        self.mres = None


def alp_bank(sup, ld, h, w, C, mask, pool_w, kernel_size, thresh=0.95, eps=1e-4, bank=None, force_mode=-1, bmask=None):
    """sup: fp32 token-major support features (row stride ld); mask fp32 [MH,MW] foreground mask."""
    _req(sup, torch.float32, "sup"); _req(mask, torch.float32, "mask")
    assert mask.dim() == 2 and mask.is_contiguous()
    ncell = (h // pool_w) * (w // pool_w)
    if bank is None:
        bank = AlpBank(ncell, C, sup.device)
    if bank.mres is None or bank.mres.numel() != 2 * h * w:
        bank.mres = torch.empty(2 * h * w, dtype=torch.float32, device=sup.device)
    if bmask is not None:
        _req(bmask, torch.float32, "bmask")
        assert bmask.shape == mask.shape and bmask.is_contiguous()
    st = _lib.lib().psam_alp_bank(_ptr(sup), ld, h, w, C, _ptr(mask), _ptr(bmask), mask.shape[0], mask.shape[1], pool_w,
                                 kernel_size,
                                 float(thresh), float(eps), _ptr(bank.bank), bank.cap, _ptr(bank.meta),
                                 _ptr(bank.slot_bg), _ptr(bank.slot_fg), _ptr(bank.mres), force_mode, _stream())
    _lib.check(st, "psam_alp_bank")
    return bank


def alp_sim(qry, q_bstride, ld, B, npix, C, bank, pred=None, part=None, eps=1e-4, sim_scale=20.0, which_only=-1):
    """qry fp32 token-major [B][npix, C] -> pred fp32 [B, 2, npix] (bg, fg)."""
    _req(qry, torch.float32, "qry")
    npt = (bank.cap + 63) // 64
    npix_pad = (npix + 63) // 64 * 64
    if part is None:
        part = torch.empty(2 * B * npt * npix_pad * 3, dtype=torch.float32, device=qry.device)
    if pred is None:
        pred = torch.empty((B, 2, npix), dtype=torch.float32, device=qry.device)
    h = _tstart("alp_sim")
    st = _lib.lib().psam_alp_sim(_ptr(qry), q_bstride, ld, B, npix, C, _ptr(bank.bank), bank.cap, _ptr(bank.meta),
                                float(eps), float(sim_scale), _ptr(part), _ptr(pred), which_only, _stream())
    # bytes: bank at capacity (an upper bound); FLOPs: every query pixel against 2 x cap prototypes on the fp32 MFMA (also an upper bound:
    # the kernel walks the occupied slots only) - this kernel's roof is the 157 TFLOP/s fp32-MFMA rate, not HBM
    _tstop(h, (B * npix * C * 4 + 2 * bank.cap * C * 4 + B * 2 * npix * 4, 2.0 * B * npix * C * 2 * bank.cap))
    _lib.check(st, "psam_alp_sim")
    return pred


# ---- resampling / packing ---------------------------------------------------------------------------------
def patchify_bilinear(img, S, P, Kpad, out=None):
    _req(img, torch.float32, "img")
    assert img.is_contiguous() and img.dim() == 4
    B, C, H, W = img.shape
    if out is None:
        out = torch.empty((B * (S // P) ** 2, Kpad), dtype=torch.float16, device=img.device)
    st = _lib.lib().psam_patchify_bilinear(_ptr(img), B, C, H, W, S, P, Kpad, _ptr(out), _stream())
    _lib.check(st, "psam_patchify_bilinear")
    return out


def bilinear_nchw(x, OH, OW, out=None):
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    planes = x.numel() // (x.shape[-1] * x.shape[-2])
    if out is None:
        out = torch.empty(tuple(x.shape[:-2]) + (OH, OW), dtype=torch.float32, device=x.device)
    st = _lib.lib().psam_bilinear_nchw(_ptr(x), planes, x.shape[-2], x.shape[-1], OH, OW, _ptr(out), _stream())
    _lib.check(st, "psam_bilinear_nchw")
    return out


def resize2d(x, OH, OW, mode, out=None):
    """F.interpolate of fp32 [..., H, W] planes: mode 0 bilinear, 1 bilinear align_corners=True, 2 nearest."""
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    planes = x.numel() // (x.shape[-1] * x.shape[-2])
    if out is None:
        out = torch.empty(tuple(x.shape[:-2]) + (OH, OW), dtype=torch.float32, device=x.device)
    st = _lib.lib().psam_resize2d(_ptr(x), planes, x.shape[-2], x.shape[-1], OH, OW, mode, _ptr(out), _stream())
    _lib.check(st, "psam_resize2d")
    return out


def prob_argmax(logits, OH, OW, prob=None, pred=None, fg_sum=None):
    _req(logits, torch.float32, "logits")
    assert logits.is_contiguous() and logits.dim() == 4 and logits.shape[1] == 2
    B = logits.shape[0]
    if prob is None:
        prob = torch.empty((B, 2, OH, OW), dtype=torch.float32, device=logits.device)
    if pred is None:
        pred = torch.empty((B, OH, OW), dtype=torch.uint8, device=logits.device)
    h = _tstart("prob_argmax")
    st = _lib.lib().psam_prob_argmax(_ptr(logits), B, logits.shape[2], logits.shape[3], OH, OW, _ptr(prob), _ptr(pred),
                                    _ptr(fg_sum), _stream())
    _tstop(h, B * (2 * logits.shape[2] * logits.shape[3] * 4 + OH * OW * 9))
    _lib.check(st, "psam_prob_argmax")
    return prob, pred


def broadcast_rows(row, out, B, stride, off):
    _req(row, torch.float32, "row"); _req(out, torch.float32, "out")
    st = _lib.lib().psam_broadcast_rows(_ptr(row), row.numel(), _ptr(out), B, stride, off, _stream())
    _lib.check(st, "psam_broadcast_rows")


def minmax(x, B, mm=None):
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    if mm is None:
        mm = torch.empty(2 * B, dtype=torch.int32, device=x.device)
    st = _lib.lib().psam_minmax(_ptr(x), B, x.numel() // B, _ptr(mm), _stream())
    _lib.check(st, "psam_minmax")
    return mm


def sam_patchify(img, mm, S, P, mean3, std3, quantise=True, out=None, u8out=None):
    import ctypes
    _req(img, torch.float32, "img")
    assert img.is_contiguous() and img.shape[-1] == S and img.shape[-2] == S
    B = img.shape[0]
    if out is None:
        out = torch.empty((B * (S // P) ** 2, 3 * P * P), dtype=torch.float16, device=img.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    s = (ctypes.c_float * 3)(*[float(v) for v in std3])
    st = _lib.lib().psam_sam_patchify(_ptr(img), _ptr(mm), B, S, P, m, s, 1 if quantise else 0, _ptr(out), _ptr(u8out),
                                     _stream())
    _lib.check(st, "psam_sam_patchify")
    return out


def im2col3x3(x, B, H, W, C, out=None):
    _req(x, torch.float16, "x")
    assert x.is_contiguous()
    if out is None:
        out = torch.empty((B * H * W, 9 * C), dtype=torch.float16, device=x.device)
    st = _lib.lib().psam_im2col3x3(_ptr(x), B, H, W, C, _ptr(out), _stream())
    _lib.check(st, "psam_im2col3x3")
    return out


def cast_f16(x, out=None):
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    st = _lib.lib().psam_cast_f16(_ptr(x), _ptr(out), x.numel(), _stream())
    _lib.check(st, "psam_cast_f16")
    return out


def cast_f32(x, out=None):
    """fp16 -> fp32 (the reference-width encoder mode: the attention output on its way to psam_gemm_f32x3)."""
    _req(x, torch.float16, "x")
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _req(out, torch.float32, "out")
    st = _lib.lib().psam_cast_f32(_ptr(x), _ptr(out), x.numel(), _stream())
    _lib.check(st, "psam_cast_f32")
    return out


def gelu_f32_(x):
    """nn.GELU (erf form) in place on fp32 (the reference-width encoder mode, between lin1 and lin2)."""
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    st = _lib.lib().psam_gelu_f32(_ptr(x), x.numel(), _stream())
    _lib.check(st, "psam_gelu_f32")
    return x


def split_f16(x, hi=None, lo=None, write_hi=True):
    """fp32 x -> (hi = half(x), lo = half(x - hi)); write_hi=False: `hi` already holds half(x) (a folded-LayerNorm GEMM wrote it)."""
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    if hi is None:
        assert write_hi
        hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    if lo is None:
        lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _req(hi, torch.float16, "hi"); _req(lo, torch.float16, "lo")
    st = _lib.lib().psam_split_f16(_ptr(x), _ptr(hi), _ptr(lo), x.numel(), 1 if write_hi else 0, _stream())
    _lib.check(st, "psam_split_f16")
    return hi, lo


def split_weight_f16(w, scale=1.0):
    """fp32 weight -> (hi, lo) fp16 pair of w * scale (one-time packing; scale a power of two: gemm_f32x3 undoes it)."""
    w = w.detach().float()
    # |w * scale| must stay inside fp16 (hi = inf, lo = -inf, NaN products otherwise): halve the scale until it does
    amax = float(w.abs().max()) if w.numel() else 0.0
    if not amax < float("inf"):
        raise ValueError("split_weight_f16: non-finite weight")
    if amax * scale >= 6.0e4:
        raise ValueError(f"split_weight_f16: max |w| = {amax:.4g} times the scale {scale:g} overflows fp16; use `split_scale_for(w)`")
    w = w * scale
    hi = w.half()
    return hi.contiguous(), (w - hi.float()).half().contiguous()


def split_scale_for(w, want=256.0):
    """the largest power of two <= `want` that keeps max |w| * scale below the fp16 range (gemm_f32x3's weight scale)"""
    amax = float(w.detach().abs().max()) if w.numel() else 0.0
    s = float(want)
    while s > 2.0 ** -14 and amax * s >= 6.0e4:
        s *= 0.5
    return s


X3_WEIGHT_SCALE = 256.0      # gemm_f32x3: weights are split as w * 2^8 (a lo half is a normal fp16 down to |w| = 5e-4; |w| < 255)


# ---- SAM prompt encoder / mask decoder -----------------------------------------------------------------------
def small_linear(x, W, b=None, out=None, act=0, resid=None, x2=None, G=1, M=None, N=None, K=None, xg=0, wg=0, bg=0, yg=0,
                 ldx=None, ldy=None):
    """Grouped fp32 y[g,m,:] = act(x[g,m,:] @ W[g]^T + b[g]) (+ resid). Defaults: one group, x [M,K], W [N,K]."""
    _req(x, torch.float32, "x"); _req(W, torch.float32, "W"); _req(b, torch.float32, "b")
    _req(resid, torch.float32, "resid")
    if M is None:
        M = x.shape[-2]
    if K is None:
        K = x.shape[-1]
    if N is None:
        N = W.shape[-2]
    if ldx is None:
        ldx = x.stride(-2)
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    if ldy is None:
        ldy = out.stride(-2)
    _req(x2, torch.float32, "x2")
    st = _lib.lib().psam_small_linear(_ptr(x), _ptr(x2), _ptr(W), _ptr(b), _ptr(resid), _ptr(out), G, M, N, K, xg, wg, bg, yg,
                                     ldx, ldy, act, _stream())
    _lib.check(st, "psam_small_linear")
    return out


def small_attention(q, k, v, out, B, Tq, Tk, NH, hd, ldq, ldk, ldv, ldo):
    q16 = q.dtype == torch.float16
    _req(k, torch.float32, "k"); _req(v, torch.float32, "v")
    assert out.dtype == q.dtype
    st = _lib.lib().psam_small_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Tq, Tk, NH, hd, ldq, ldk, ldv, ldo,
                                        1 if q16 else 0, _stream())
    _lib.check(st, "psam_small_attention")
    return out


def t2i_split(B, NH, T, Nk, n_cu=256):
    """into how many key ranges a token-to-image attention launch is split (PSAM_T2I_SPLIT: "auto", or a number to force): the unsplit
    launch has B * NH workgroups whose waves each walk all keys, 64 dependent round trips"""
    mode = _os.environ.get("PSAM_T2I_SPLIT", "auto")
    if T > 16 or Nk < 512 or mode in ("0", "1"):        # (a forced count of 1 is one range: no split)
        return 1
    if mode != "auto":
        return max(1, min(16, Nk // 256, int(mode)))
    return max(1, min(16, Nk // 256, n_cu // (B * NH)))


def t2i_attention(q, K, V, out, B, T, Nk, NH, head_major=False, split=None):
    """head_major: K / V are [B][NH][Nk][16] (gemm_f32(..., heads=(Nk, 16))) instead of token-major [B][Nk][NH * 16].
    split = (S, part): the keys in S ranges per (prompt set, head), merged by a second small launch - part fp32 scratch of at least
    B * NH * S * T * 18 elements."""
    _req(q, torch.float32, "q"); _req(K, K.dtype, "K"); _req(V, K.dtype, "V"); _req(out, torch.float32, "out")
    assert K.dtype in (torch.float16, torch.float32)
    if q.shape[-1] != NH * 16:     # the kernels (and the head-major K / V layout) are written for 16-wide heads: C = NH * 16
        raise ValueError(f"psam_t2i_attention: head size {q.shape[-1]}/{NH} is not 16 (SAM: 128 channels, 8 heads)")
    flags = (1 if K.dtype == torch.float32 else 0) | (2 if head_major else 0)
    if split is not None and split[0] > 1:
        S, part = split
        _req(part, torch.float32, "part")
        assert part.numel() >= B * NH * S * T * 18
        st = _lib.lib().psam_t2i_attention_split(_ptr(q), _ptr(K), _ptr(V), _ptr(out), B, T, Nk, NH, flags, S, _ptr(part), _stream())
        _lib.check(st, "psam_t2i_attention_split")
        return out
    st = _lib.lib().psam_t2i_attention(_ptr(q), _ptr(K), _ptr(V), _ptr(out), B, T, Nk, NH, flags, _stream())
    _lib.check(st, "psam_t2i_attention")
    return out


def small_linear_splitk(x, W, b, resid, out, parts, ks):
    """out[M,N] = x[M,K] @ W[N,K]^T + b (+ resid) as `ks` K ranges in one launch + a fixed-order sum (few rows, long contraction);
    parts: fp32 scratch of at least ks * M * N elements."""
    _req(x, torch.float32, "x"); _req(W, torch.float32, "W"); _req(b, torch.float32, "b"); _req(resid, torch.float32, "resid")
    _req(out, torch.float32, "out"); _req(parts, torch.float32, "parts")
    M, K = x.shape[-2], x.shape[-1]
    N = W.shape[0]
    assert W.shape[1] == K and W.is_contiguous() and parts.numel() >= ks * M * N and out.stride(-1) == 1
    assert resid is None or resid.stride(-2) == out.stride(-2)
    st = _lib.lib().psam_small_linear_splitk(_ptr(x), _ptr(W), _ptr(b), _ptr(resid), _ptr(out), _ptr(parts), M, N, K, ks, x.stride(-2),
                                            out.stride(-2), _stream())
    _lib.check(st, "psam_small_linear_splitk")
    return out


def gemm_f32(a, w, bias=None, out=None, resid=None, a2=None, a2_mod=0, heads=None):
    """out[M,N] = (a[M,K] [+ a2[m % a2_mod]]) @ w[N,K]^T + bias [+ resid]; everything fp32 (exact-fp32 MFMA).
    heads = (nk, hd): the result is written head-major, out fp32 [M / nk][N / hd][nk][hd] (no residual)."""
    if heads is not None:
        _req(a, torch.float32, "a"); _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias"); _req(a2, torch.float32, "a2")
        nk, hd = heads
        M, K = a.shape
        N = w.shape[0]
        assert resid is None and out is not None and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= M * N
        assert a2 is None or (a2.stride(0) == a.stride(0) and a2_mod > 0)
        st = _lib.lib().psam_gemm_f32_heads(_ptr(a), _ptr(a2), a2_mod, _ptr(w), _ptr(bias), _ptr(out), M, N, K, a.stride(0), w.stride(0),
                                           nk, hd, _stream())
        _lib.check(st, "psam_gemm_f32_heads")
        return out
    _req(a, torch.float32, "a"); _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias")
    _req(resid, torch.float32, "resid"); _req(a2, torch.float32, "a2")
    assert a.dim() == 2 and w.dim() == 2 and w.shape[1] == a.shape[1]
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _req(out, torch.float32, "out")
    assert out.shape == (M, N) and (resid is None or (resid.shape == (M, N) and resid.stride(0) == out.stride(0)))
    assert a2 is None or (a2.stride(0) == a.stride(0) and a2_mod > 0)
    st = _lib.lib().psam_gemm_f32(_ptr(a), _ptr(a2), a2_mod, _ptr(w), _ptr(bias), _ptr(resid), _ptr(out), M, N, K,
                                 a.stride(0), w.stride(0), out.stride(0), _stream())
    _lib.check(st, "psam_gemm_f32")
    return out


def gemm_f32x3(a, w_split, bias=None, out=None, resid=None, a2=None, a2_mod=0, heads=None):
    """gemm_f32 at fp32 accuracy on the fp16 matrix pipe: w_split = split_weight_f16(w) - the (hi, lo) fp16 halves of the fp32 weight;
    a (fp32) is split on chip, three MFMA products per k step, fp32 accumulation (psam_gemm_f32x3). heads = (nk, hd) as gemm_f32.
    w_split may carry a third element: the power of two the weight was scaled by before the split (default 1)."""
    wh, wl = w_split[0], w_split[1]
    wscale = float(w_split[2]) if len(w_split) > 2 else 1.0
    _req(a, torch.float32, "a"); _req(wh, torch.float16, "wh"); _req(wl, torch.float16, "wl"); _req(bias, torch.float32, "bias")
    _req(resid, torch.float32, "resid"); _req(a2, torch.float32, "a2")
    assert a.dim() == 2 and wh.dim() == 2 and wh.shape == wl.shape and wh.shape[1] == a.shape[1] and wh.stride(0) == wl.stride(0)
    M, K = a.shape
    N = wh.shape[0]
    assert a2 is None or (a2.stride(0) == a.stride(0) and a2_mod > 0)
    nk, hd = heads if heads is not None else (0, 0)
    if heads is not None:
        assert resid is None and out is not None and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= M * N
        ldo = N
    else:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _req(out, torch.float32, "out")
        assert out.shape == (M, N) and (resid is None or (resid.shape == (M, N) and resid.stride(0) == out.stride(0)))
        ldo = out.stride(0)
    st = _lib.lib().psam_gemm_f32x3(_ptr(a), _ptr(a2), a2_mod, _ptr(wh), _ptr(wl), _ptr(bias), _ptr(resid), _ptr(out), M, N, K,
                                   a.stride(0), wh.stride(0), ldo, nk, hd, 1.0 / wscale, _stream())
    _lib.check(st, "psam_gemm_f32x3")
    return out


def ln_pe(x, pe, M, y32=None, y16=None, ype16=None, add_vec=None, w=None, b=None, in_mod=0, pe_mod=4096, eps=1e-5,
          img_of_prompt=None):
    _req(x, torch.float32, "x"); _req(pe, torch.float32, "pe"); _req(img_of_prompt, torch.int32, "img_of_prompt")
    st = _lib.lib().psam_ln_pe(_ptr(x), _ptr(add_vec), _ptr(w), _ptr(b), _ptr(pe), _ptr(y32), _ptr(y16), _ptr(ype16), M,
                              in_mod, pe_mod, float(eps), 1 if w is not None else 0, _ptr(img_of_prompt), _stream())
    _lib.check(st, "psam_ln_pe")


def dense_pe(G, gh, gw):
    _req(G, torch.float32, "G")
    pe = torch.empty((gh * gw, 256), dtype=torch.float32, device=G.device)
    st = _lib.lib().psam_dense_pe(_ptr(G), gh, gw, _ptr(pe), _stream())
    _lib.check(st, "psam_dense_pe")
    return pe


def prompt_tokens(coords, labels, G, type_emb, out_tok, B, Ns, img_size, tokens=None):
    _req(coords, torch.float32, "coords"); _req(labels, torch.int32, "labels")
    if tokens is None:
        tokens = torch.empty((B, 5 + Ns, 256), dtype=torch.float32, device=G.device)
    st = _lib.lib().psam_prompt_tokens(_ptr(coords), _ptr(labels), _ptr(G), _ptr(type_emb), _ptr(out_tok), B, Ns,
                                      float(img_size), _ptr(tokens), _stream())
    _lib.check(st, "psam_prompt_tokens")
    return tokens


def upscale_tail(u1, lnw, lnb, W2r, b2, hyper, B, g, masks=None):
    _req(u1, torch.float32, "u1"); _req(hyper, torch.float32, "hyper")
    if masks is None:
        masks = torch.empty((B, 4, 4 * g, 4 * g), dtype=torch.float32, device=u1.device)
    st = _lib.lib().psam_upscale_tail(_ptr(u1), _ptr(lnw), _ptr(lnb), _ptr(W2r), _ptr(b2), _ptr(hyper), _ptr(masks), B, g,
                                     _stream())
    _lib.check(st, "psam_upscale_tail")
    return masks


def mask_upsample(low, MID, variant, out=None):
    _req(low, torch.float32, "low")
    assert low.is_contiguous()
    IN = low.shape[-1]
    planes = low.numel() // (IN * IN)
    if out is None:
        out = torch.empty(tuple(low.shape[:-2]) + (MID, MID), dtype=torch.float32, device=low.device)
    st = _lib.lib().psam_mask_upsample(_ptr(low), planes, IN, MID, variant, _ptr(out), _stream())
    _lib.check(st, "psam_mask_upsample")
    return out


def mask_union(low, sel, MID, OUT, variant, thr=0.0, pred=None):
    _req(low, torch.float32, "low")
    assert low.is_contiguous() and low.dim() == 4
    B, C, IN, _ = low.shape
    if pred is None:
        pred = torch.empty((OUT, OUT), dtype=torch.float32, device=low.device)
    st = _lib.lib().psam_mask_union(_ptr(low), B, C, sel, IN, MID, OUT, variant, float(thr), _ptr(pred), _stream())
    _lib.check(st, "psam_mask_union")
    return pred


def mask_downscale(masks, wts, g, eps=1e-6, out=None):
    """PromptEncoder.mask_downscaling: masks fp32 [n,4g,4g] -> dense embeddings fp32 [n, g*g, 256] (token-major)."""
    _req(masks, torch.float32, "masks"); _req(wts, torch.float32, "wts")
    assert masks.is_contiguous() and masks.shape[-1] == 4 * g and masks.shape[-2] == 4 * g and wts.numel() == 4684
    n = masks.numel() // (16 * g * g)
    if out is None:
        out = torch.empty((n, g * g, 256), dtype=torch.float32, device=masks.device)
    st = _lib.lib().psam_mask_downscale(_ptr(masks), _ptr(wts), n, g, float(eps), _ptr(out), _stream())
    _lib.check(st, "psam_mask_downscale")
    return out


def mask_stats(low, first, nsel, MID, H, W, variant, thr=0.0, off=1.0, stats=None):
    """int32 [B*nsel, 8] = {n(v>thr+off), n(v>thr-off), n(v>thr), min_x, min_y, max_x, max_y, 0} per selected plane of
    low [B,C,IN,IN] over the [H,W] corner of its MID x MID up-sampling."""
    _req(low, torch.float32, "low")
    assert low.is_contiguous() and low.dim() == 4
    B, C, IN, _ = low.shape
    if stats is None:
        stats = torch.empty((B * nsel, 8), dtype=torch.int32, device=low.device)
    st = _lib.lib().psam_mask_stats(_ptr(low), B, C, first, nsel, IN, MID, H, W, variant, float(thr), float(off),
                                   _ptr(stats), _stream())
    _lib.check(st, "psam_mask_stats")
    return stats


def plane_stats(planes, thr=0.0, off=1.0, binarize=True):
    """planes fp32 [n,H,W] -> (stats int32 [n,8] as mask_stats, uint8 [n,H,W] = plane > thr or None)."""
    _req(planes, torch.float32, "planes")
    assert planes.is_contiguous() and planes.dim() == 3
    n, H, W = planes.shape
    stats = torch.empty((n, 8), dtype=torch.int32, device=planes.device)
    out = torch.empty((n, H, W), dtype=torch.uint8, device=planes.device) if binarize else None
    st = _lib.lib().psam_plane_stats(_ptr(planes), n, H, W, float(thr), float(off), _ptr(stats), _ptr(out), _stream())
    _lib.check(st, "psam_plane_stats")
    return stats, out


def mask_binarize(low, idx, MID, H, W, variant, thr=0.0, label=None, out=None):
    """uint8 [n,H,W] = upsample(low.view(-1,IN,IN)[idx[i]]) > thr; with `label` uint8 [H,W] also int64 [n,3] {tp,fp,fn}."""
    _req(low, torch.float32, "low"); _req(idx, torch.int32, "idx")
    assert low.is_contiguous() and idx.is_contiguous()
    IN, n = low.shape[-1], idx.numel()
    if out is None:
        out = torch.empty((n, H, W), dtype=torch.uint8, device=low.device)
    counts = None
    if label is not None:
        _req(label, torch.uint8, "label")
        assert label.is_contiguous() and tuple(label.shape) == (H, W)
        counts = torch.empty((n, 3), dtype=torch.int64, device=low.device)
    st = _lib.lib().psam_mask_binarize(_ptr(low), _ptr(idx), n, IN, MID, H, W, variant, float(thr), _ptr(out),
                                      _ptr(label), _ptr(counts), _stream())
    _lib.check(st, "psam_mask_binarize")
    return out, counts


def normalize_chw(x, mean3, std3, out=None):
    """(x - mean[c]) / std[c] on [B,3,H,W]; x uint8 or fp32."""
    import ctypes
    assert x.is_cuda and x.is_contiguous() and x.dim() == 4 and x.shape[1] == 3
    assert x.dtype in (torch.uint8, torch.float32)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    s = (ctypes.c_float * 3)(*[float(v) for v in std3])
    st = _lib.lib().psam_normalize_chw(_ptr(x), 1 if x.dtype == torch.uint8 else 0, x.shape[0],
                                      x.shape[2] * x.shape[3], m, s, _ptr(out), _stream())
    _lib.check(st, "psam_normalize_chw")
    return out



# ---- HIP graph replay of a launch-only forward -------------------------------------------------------------------
def graph_wanted(x, max_batch_rows):
    """PSAM_HIPGRAPH = auto (default: small calls only) / 1 / 0. Never while per-kernel timers are attached (they need the launches
    themselves) or inside another capture."""
    import os
    mode = os.environ.get("PSAM_HIPGRAPH", "auto")
    if mode == "0" or not x.is_cuda or torch.cuda.is_current_stream_capturing():
        return False
    if TIMERS or GEMM_TIMER is not None:
        return False
    return mode == "1" or x.shape[0] <= max_batch_rows


class GraphCache:
    """fn(static_input) -> output living in a persistent workspace, captured once per key and replayed: one host call instead of the
    forward's hundred-odd launches. A failed capture (a runtime that cannot record these launches) falls back to eager, once."""

    def __init__(self, what, limit=4):
        self.what, self.limit, self.graphs = what, limit, {}

    def run(self, key, x, fn):
        """x: a tensor or a tuple of tensors (the inputs that change from call to call: copied into static clones); fn(*statics)."""
        xs = x if isinstance(x, (tuple, list)) else (x,)
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= self.limit:
                self.graphs.clear()
            statics = tuple(t.contiguous().clone() for t in xs)
            try:
                # warm-up outside the capture (work lists, workspaces and weight packs are built on first use), then the capture
                dev = xs[0].device
                cur = torch.cuda.current_stream(dev)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fn(*statics)
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = fn(*statics)
                ent = self.graphs[key] = (g, statics, out)
            except Exception as e:
                import sys
                ent = self.graphs[key] = (None, None, None)
                print(f"protosam_amd: HIP graph capture of {self.what} failed ({e!r}); running eagerly", file=sys.stderr)
        g, statics, out = ent
        if g is None:
            return None
        for st, t in zip(statics, xs):
            st.copy_(t)
        g.replay()
        return out

# ---- connected components --------------------------------------------------------------------------------------
CC_HDR, CC_STRIDE = 8, 12


class CclWorkspace:
    """Scratch for psam_ccl plus `slots` result tables (one per image of a batch), and their pinned host mirror."""

    def __init__(self, H, W, cap, device, slots=1):
        n = H * W
        self.H, self.W, self.cap, self.slots = H, W, cap, slots
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=device)  # noqa: E731
        # every scratch array `slots` times (ccl_batch works on all images of a batch at once; ccl on image 0's part)
        self.parent_b, self.labels_b = i32(slots * n).view(slots, n), i32(slots * n).view(slots, n)
        self.counters_b, self.roots_b, self.acc_i_b = i32(slots * 2), i32(slots * cap), i32(slots * 5 * cap)
        self.acc_u_b = torch.empty(slots * 3 * cap, dtype=torch.int64, device=device)
        self.acc_d_b = torch.empty(slots * cap, dtype=torch.float64, device=device)
        self.parent, self.labels, self.counters, self.roots = self.parent_b[0], self.labels_b[0], self.counters_b[:2], self.roots_b[:cap]
        self.acc_i, self.acc_u, self.acc_d = self.acc_i_b[:5 * cap], self.acc_u_b[:3 * cap], self.acc_d_b[:cap]
        self.tabs = torch.zeros((slots, CC_HDR + CC_STRIDE * cap), dtype=torch.float64, device=device)
        self.tabs_host = torch.empty((slots, CC_HDR + CC_STRIDE * cap), dtype=torch.float64).pin_memory()

    @property
    def tab(self):
        return self.tabs[0]

    @property
    def tab_host(self):
        return self.tabs_host[0]


def ccl(pred_u8, pfg, ws, fg_sum=None, slot=0):
    """pred uint8 [H,W], pfg fp32 [H,W] -> ws.labels (int32 [H*W]) and ws.tabs[slot] (fp64 table, see csrc/ccl.hip)."""
    assert pred_u8.dtype == torch.uint8 and pred_u8.is_cuda and pred_u8.is_contiguous()
    _req(pfg, torch.float32, "pfg")
    h = _tstart("ccl")
    st = _lib.lib().psam_ccl(_ptr(pred_u8), _ptr(pfg), ws.H, ws.W, ws.cap, _ptr(ws.labels), _ptr(ws.parent),
                            _ptr(ws.counters), _ptr(ws.roots), _ptr(ws.acc_i), _ptr(ws.acc_u), _ptr(ws.acc_d),
                            _ptr(fg_sum), _ptr(ws.tabs[slot]), _stream())
    _tstop(h, ws.H * ws.W * (1 + 4 + 4))      # pred u8 + p_fg fp32 in, labels int32 out (the table is a few KB)
    _lib.check(st, "psam_ccl")
    return ws


def ccl_batch(pred_u8, prob, ws, fg_sum=None):
    """pred uint8 [B,H,W], prob fp32 [B,2,H,W] (foreground = channel 1) -> ws.labels_b[b], ws.tabs[b] for every image in ONE chain of
    six launches (psam_ccl_batch)."""
    B = pred_u8.shape[0]
    assert pred_u8.dtype == torch.uint8 and pred_u8.is_cuda and pred_u8.is_contiguous() and B <= ws.slots
    _req(prob, torch.float32, "prob")
    assert prob.is_contiguous() and prob.shape[1] == 2
    h = _tstart("ccl")
    st = _lib.lib().psam_ccl_batch(_ptr(pred_u8), _ptr(prob[0, 1]), 2 * ws.H * ws.W, B, ws.H, ws.W, ws.cap, _ptr(ws.labels_b),
                                  _ptr(ws.parent_b), _ptr(ws.counters_b), _ptr(ws.roots_b), _ptr(ws.acc_i_b), _ptr(ws.acc_u_b),
                                  _ptr(ws.acc_d_b), _ptr(fg_sum), _ptr(ws.tabs), _stream())
    _tstop(h, B * ws.H * ws.W * (1 + 4 + 4))
    _lib.check(st, "psam_ccl_batch")
    return ws


def neg_points(ws, pbg, tab, max_comp, r=10, thr=0.95, keys=None, labels=None):
    """int64 keys [max_comp+1] (see psam_neg_points) from ws.labels (the CCL of the SAME image) and p_bg fp32 [H,W]."""
    _req(pbg, torch.float32, "pbg")
    assert pbg.is_contiguous() and tab.dtype == torch.float64
    if keys is None:
        keys = torch.empty(max_comp + 1, dtype=torch.int64, device=pbg.device)
    st = _lib.lib().psam_neg_points(_ptr(ws.labels if labels is None else labels), _ptr(pbg), _ptr(tab), ws.H, ws.W, max_comp, r,
                                   float(thr), _ptr(keys), _stream())
    _lib.check(st, "psam_neg_points")
    return keys


def decode_point_key(key, W):
    """key (python int, possibly negative from the int64 view) -> (x, y, p) or None."""
    key &= (1 << 64) - 1
    if key == 0:
        return None
    import struct
    idx = 0xFFFFFFFF - (key & 0xFFFFFFFF)
    return idx % W, idx // W, struct.unpack("<f", struct.pack("<I", key >> 32))[0]


def volume_stats(vol, vol_dtype, slope=1.0, inter=0.0):
    """fp64 [2] = {sum(x), sum(x^2)} of x = voxel*slope + inter over a raw device volume (psam_volume_stats)."""
    assert vol.is_cuda and vol.is_contiguous()
    out = torch.empty(2, dtype=torch.float64, device=vol.device)
    st = _lib.lib().psam_volume_stats(_ptr(vol), vol_dtype, vol.numel(), float(slope), float(inter), _ptr(out), _stream())
    _lib.check(st, "psam_volume_stats")
    return out


def volume_slices(vol, vol_dtype, Z, H, W, slope, inter, mean, inv_std, S, tile, mode, out=None):
    """fp32 [Z,tile,S,S]: normalise + cv2-style resize (mode 0 linear, 1 nearest) + channel tiling (psam_volume_slices)."""
    assert vol.is_cuda and vol.is_contiguous()
    if out is None:
        out = torch.empty((Z, tile, S, S), dtype=torch.float32, device=vol.device)
    st = _lib.lib().psam_volume_slices(_ptr(vol), vol_dtype, Z, H, W, float(slope), float(inter), float(mean), float(inv_std),
                                      S, tile, mode, _ptr(out), _stream())
    _lib.check(st, "psam_volume_slices")
    return out


def bilinear_tokens(tok, in_bstride, ld, B, ih, iw, C, oh, ow, out=None):
    """fp32 token-major [B][ih*iw, C] (strided) -> contiguous [B, oh*ow, C]."""
    _req(tok, torch.float32, "tok")
    if out is None:
        out = torch.empty((B, oh * ow, C), dtype=torch.float32, device=tok.device)
    st = _lib.lib().psam_bilinear_tokens(_ptr(tok), in_bstride, ld, B, ih, iw, C, oh, ow, _ptr(out), _stream())
    _lib.check(st, "psam_bilinear_tokens")
    return out
