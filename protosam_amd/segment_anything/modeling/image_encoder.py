"""SAM ViTDet image encoder on the HIP kernels (state-dict names of the vendored reference module).

Mirrors /root/reference/models/segment_anything/modeling/image_encoder.py: `ImageEncoderViT` (:17-122), `Block`
(:125-193), `Attention` (:196-251), windowing (:254-300), decomposed rel-pos (:303-372), `PatchEmbed` (:375-406).
Per block: LN -> QKV GEMM -> psam_relpos (from the unscaled fp16 q) -> fused attention (14x14 windows folded into
index math, or global with a 64-key tile = one key row) -> proj GEMM (+residual) -> LN -> lin1 GEMM (+GELU) ->
lin2 GEMM (+residual). Neck: 1x1 conv = GEMM, LayerNorm2d = row LN (token-major), 3x3 conv = im2col + GEMM, LN.
The 64x64x256 embedding is kept token-major; `forward` returns the reference's NCHW shape as a permuted view.
"""
import os

import torch
import torch.nn as nn

from ... import ops
from .common import LayerNorm2d, MLPBlock, f16, f32

LN_EPS = 1e-6  # build_sam.py:72


class PatchEmbed(nn.Module):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)

    @torch.no_grad()
    def forward(self, x):
        """image_encoder.py:402-406: x [B,3,H,W] -> [B,H/P,W/P,embed_dim] (conv with stride = kernel as im2col + `psam_gemm_f16`;
        square images and a square, unpadded kernel - SAM's configuration)."""
        B, C, H, W = x.shape
        P = self.proj.kernel_size[0]
        if self.proj.kernel_size != (P, P) or self.proj.stride != (P, P) or self.proj.padding != (0, 0) or H != W or H % P:
            raise NotImplementedError("PatchEmbed.forward: square image, square kernel = stride, no padding (build_sam.py:66-81)")
        D = self.proj.out_channels
        K = C * P * P
        patches = ops.patchify_bilinear(x.float().contiguous(), H, P, (K + 63) // 64 * 64)       # (same size: the identity resample)
        w = torch.zeros((D, patches.shape[1]), dtype=torch.float16, device=x.device)
        w[:, :K] = self.proj.weight.detach().reshape(D, K).half()
        out = ops.gemm(patches, w, f32(self.proj.bias), epilogue=ops.EPI_F32)
        return out.view(B, H // P, W // P, D)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True, input_size=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))

    def _attend(self, qkv, B, g, ws, rpack, pad_row, out=None, relq=None, rel_bufs=None):
        """The attention of one block on the packed projection qkv fp16 [B*g*g, 3*dim] of a g x g token map: ws > 0 the 14 x 14 windows
        (zero padding and un-partition as index math, image_encoder.py:254-300), else global; decomposed rel-pos inside the kernels
        where they compute it themselves, through `psam_relpos` elsewhere (:303-372). -> fp16 [B*g*g, dim]."""
        H = self.num_heads
        N = g * g
        hd = qkv.shape[-1] // (3 * H)
        hm = ops.QKV_HEAD_MAJOR
        if ws > 0:
            if ops.FUSE_WINDOW_RELPOS:   # the query-side rel-pos terms are computed inside the attention kernel
                return ops.attention(qkv, B, N, H, hd, self.scale, out=out, mode=2, rpack=rpack, pad_row=pad_row, gh=g, gw=g, ws=ws,
                                     head_major=hm)
            relq = ops.relpos(qkv, rpack, B, N, H, hd, g, ws, True, self.scale, relq=relq, head_major=hm)
            return ops.attention(qkv, B, N, H, hd, self.scale, out=out, mode=2, relq=relq, pad_row=pad_row, gh=g, gw=g, ws=ws,
                                 head_major=hm)
        if ops.attention_fused_relpos(B, N, H, hd, g, g):   # rel_h / rel_w computed inside the attention kernel (round 5)
            return ops.attention(qkv, B, N, H, hd, self.scale, out=out, mode=1, rpack=rpack, gh=g, gw=g)
        relh, relw = rel_bufs() if rel_bufs is not None else (None, None)
        relh, relw = ops.relpos(qkv, rpack, B, N, H, hd, g, g, False, self.scale, rel_h=relh, rel_w=relw, head_major=hm)
        return ops.attention(qkv, B, N, H, hd, self.scale, out=out, mode=1, rel_h=relh, rel_w=relw, gh=g, gw=g, head_major=hm)

    @torch.no_grad()
    def forward(self, x):
        """image_encoder.py:235-251 for the map a GLOBAL block hands over: x [B,64,64,dim] -> [B,64,64,dim] fp32 (qkv GEMM, fused
        attention with the decomposed rel-pos of `rel_pos_h / rel_pos_w`, proj GEMM). The 14 x 14 windows of a windowed block never
        exist as tensors here (`Block.forward` runs them as index math inside the kernel), so a [B*nW,14,14,dim] input has no kernel."""
        B, gh, gw, D = x.shape
        if not self.use_rel_pos or gh != gw or gh != 64 or self.rel_pos_h.shape[0] != 2 * gh - 1:
            raise NotImplementedError("Attention.forward: the 64 x 64 token map of a global block with its 127-row rel-pos tables; "
                                      "windowed blocks run through Block.forward")
        x16 = x.reshape(B * gh * gw, D).half().contiguous()
        hd = D // self.num_heads
        rpack = ops.pack_rel_tables(_rel_table(self.rel_pos_h, gh), _rel_table(self.rel_pos_w, gw), False, hd)
        if ops.QKV_HEAD_MAJOR:
            qkv = ops.gemm_heads(x16, f16(self.qkv.weight), f32(self.qkv.bias), hd)
        else:
            qkv = ops.gemm(x16, f16(self.qkv.weight), f32(self.qkv.bias), epilogue=ops.EPI_F16)
        att = self._attend(qkv, B, gh, 0, rpack, None)
        out = ops.gemm(att.view(B * gh * gw, D), f16(self.proj.weight), f32(self.proj.bias), epilogue=ops.EPI_F32)
        return out.view(B, gh, gw, D)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, input_size=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size
        self._pk = None

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load
        self._pk = None
        return super()._load_from_state_dict(*a, **k)

    def _packed(self, grid):
        """One-time weight packing of the block for a grid x grid token map (fp16 GEMM operands, fp32 biases / LayerNorm parameters,
        the rel-pos table pack, and the LayerNorm-folded forms of qkv / lin1 for `ImageEncoderViT.fold_ln`)."""
        if self._pk is None or self._pk["grid"] != grid:
            a = self.attn
            D = a.qkv.in_features
            K = self.window_size or grid
            d = dict(grid=grid, ws=self.window_size, n1w=f32(self.norm1.weight), n1b=f32(self.norm1.bias), n2w=f32(self.norm2.weight),
                     n2b=f32(self.norm2.bias), qkv_w=f16(a.qkv.weight), qkv_b=f32(a.qkv.bias), pad_row=f16(a.qkv.bias),
                     proj_w=f16(a.proj.weight), proj_b=f32(a.proj.bias),
                     rpack=ops.pack_rel_tables(_rel_table(a.rel_pos_h, K), _rel_table(a.rel_pos_w, K), self.window_size > 0,
                                               D // a.num_heads),
                     l1w=f16(self.mlp.lin1.weight), l1b=f32(self.mlp.lin1.bias), l2w=f16(self.mlp.lin2.weight),
                     l2b=f32(self.mlp.lin2.bias))
            # LayerNorm folded into the consuming GEMM (ops.fold_layernorm): W' = half(W * ln_w), row sums of W', bias + W . ln_b
            d["qkv_wf"], d["qkv_s"], d["qkv_t"] = ops.fold_layernorm(a.qkv.weight, a.qkv.bias, self.norm1.weight, self.norm1.bias)
            d["l1wf"], d["l1s"], d["l1t"] = ops.fold_layernorm(self.mlp.lin1.weight, self.mlp.lin1.bias, self.norm2.weight,
                                                               self.norm2.bias)
            self._pk = d
        return self._pk

    @torch.no_grad()
    def forward(self, x):
        """image_encoder.py:174-193: x [B,64,64,dim] -> [B,64,64,dim] fp32: x + attn(norm1(x)) (14 x 14 windows with zero padding when
        window_size > 0), then x + mlp(norm2(x)). The same launches as one block of `ImageEncoderViT._encode_patches` with the LayerNorms
        as passes of their own."""
        B, gh, gw, D = x.shape
        if gh != gw or gh != 64 or not isinstance(self.mlp.act, nn.GELU):
            raise NotImplementedError("Block.forward: SAM's 64 x 64 token map and GELU MLP (build_sam.py:66-81)")
        bp = self._packed(gh)
        M = B * gh * gw
        hd = D // self.attn.num_heads
        xr = x.reshape(M, D).float().clone()                                  # the residual stream (the input stays untouched)
        ln = ops.layernorm(xr, bp["n1w"], bp["n1b"], self.norm1.eps)
        if ops.QKV_HEAD_MAJOR:
            qkv = ops.gemm_heads(ln, bp["qkv_w"], bp["qkv_b"], hd)
        else:
            qkv = ops.gemm(ln, bp["qkv_w"], bp["qkv_b"], epilogue=ops.EPI_F16)
        att = self.attn._attend(qkv, B, gh, bp["ws"], bp["rpack"], bp["pad_row"])
        ops.gemm(att.view(M, D), bp["proj_w"], bp["proj_b"], out=xr, epilogue=ops.EPI_F32, resid=xr)
        ops.layernorm(xr, bp["n2w"], bp["n2b"], self.norm2.eps, out=ln)
        hid = ops.gemm(ln, bp["l1w"], bp["l1b"], epilogue=ops.EPI_GELU_F16)
        ops.gemm(hid, bp["l2w"], bp["l2b"], out=xr, epilogue=ops.EPI_F32, resid=xr)
        return xr.view(B, gh, gw, D)


def _rel_table(rel_pos, K):
    """`get_rel_pos` (image_encoder.py:303-334) for a square K x K attention region: a table whose length is not 2K - 1 (a
    checkpoint trained at another resolution) is resampled linearly to 2K - 1 rows, once, when the weights are packed; with
    q_size == k_size the gather index is qy - ky + K - 1, which is how the kernels address the packed table."""
    L = 2 * K - 1
    r = rel_pos.detach().float()
    if r.shape[0] == L:
        return r
    r = torch.nn.functional.interpolate(r.reshape(1, r.shape[0], -1).permute(0, 2, 1), size=L, mode="linear")
    return r.reshape(-1, L).permute(1, 0).contiguous()


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 out_chans=256, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU, use_abs_pos=True,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, global_attn_indexes=(),
                 use_grad_checkpointing=False):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.out_chans = out_chans
        self.grid = img_size // patch_size
        if not (use_abs_pos and use_rel_pos and qkv_bias):
            raise NotImplementedError("only SAM's configuration (abs pos + rel pos + qkv bias, build_sam.py:66-81)")
        if self.grid != 64 or (embed_dim // num_heads) not in (64, 80):
            raise NotImplementedError("HIP attention is built for the 64x64 token map and head dims 64 / 80")
        self.patch_embed = PatchEmbed((patch_size, patch_size), (patch_size, patch_size), in_chans=in_chans,
                                      embed_dim=embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.grid, self.grid, embed_dim))
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer, act_layer, use_rel_pos, rel_pos_zero_init,
                  window_size if i not in global_attn_indexes else 0, (self.grid, self.grid)) for i in range(depth)])
        self.neck = nn.Sequential(nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
                                  nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
                                  LayerNorm2d(out_chans))
        self._packed = None
        self._ws = {}
        # LayerNorm folded into the GEMMs either side of it (ops.gemm ... out16 / stats / ln_mr / ln_s): on by default since the
        # assembly GEMM has the folded epilogues (round 3: 145.3 vs 142.8 slices/s, same box, interleaved; the LayerNorm passes of a
        # 16-slice step cost 6 ms, the fp16 copy + row sums in the producers' epilogues and the rank-1 correction MFMAs in the
        # consumers' 3.8 ms; it also lowers the embedding error). Round 2 measured the opposite on the HIP kernels (122.2 vs 124.3:
        # +10 ms of epilogue). PSAM_FOLD_LN=0 / `fold_ln = False` selects the separate passes.
        self.fold_ln = os.environ.get("PSAM_FOLD_LN", "1") != "0"
        self.fold_min_fill = float(os.environ.get("PSAM_FOLD_MIN_FILL", "0.8"))      # ... where the launches fill the CUs (ops.fold_pays); 0 = always
        # Split-fp16 operands (round 5) where the fp16 rounding of an operand costs the most output accuracy per FLOP: the neck's two
        # GEMMs run on (hi, lo) pairs of activations AND weights (hi W_hi + lo W_hi + hi W_lo: ~22 mantissa bits of each), the patch
        # embedding on the EXACT uint8 pixel values (the SAM normalisation folded into the weights, which are split) when the caller
        # hands the quantised image over (ProtoSAM's hand-off does). tools/emulate_ln_fusion.py priced these three GEMMs at 4.0e-4 of
        # the 6.4e-4 embedding error; they are 2.5 % of the encoder's FLOPs. PSAM_SPLIT_FP16=0: plain fp16 operands (A/B).
        self.split_fp16 = os.environ.get("PSAM_SPLIT_FP16", "0") != "0"
        self._split_parts = os.environ.get("PSAM_SPLIT_PARTS", "neck,patch").split(",")     # (A/B of the two halves)
        # Reference-width mode (round 6; PSAM_ENCODER_X3=1 / `gemm_x3 = True`): EVERY Linear of the blocks at fp32 accuracy - operands
        # AND results fp32, three fp16 MFMA products on (hi, lo) halves (ops.gemm_f32x3, the decoder's image-side kernel), LayerNorm and
        # GELU as fp32 passes, the neck and the patch embedding on their split forms; only the attention products (QK^T, PV) keep fp16
        # operands. Not a throughput path (3x the matrix work on a kernel tuned for the decoder's shapes, fp32 activations in HBM): it
        # is the "same arithmetic width as the reference" line of bench.py beside the fp16-operand headline, with its own parity figure.
        self.gemm_x3 = os.environ.get("PSAM_ENCODER_X3", "0") != "0"
        # One image at a time (the reference-shaped call), mlp.lin2 as K ranges of the assembly tile + one reduce pass that also applies the
        # NEXT block's norm1 (ops.gemm_splitk_ln, round 6; caller-owned workspace: safe inside the captured graph). Measured per block of
        # ViT-H: 74.6 -> 71.4 us (tools/r06/splitk_probe.py) - 240 workgroups instead of 160 bring every CU under the power cap and the
        # clock drops from ~2.1 to ~1.7 GHz, so the three-fold shorter K loops buy 1 % of a one-slice call. Opt-in (PSAM_SPLITK_LIN2=1):
        # it changes the summation order of the one-slice path for that 1 %.
        self.splitk_lin2 = os.environ.get("PSAM_SPLITK_LIN2", "0") != "0"

    def _apply(self, fn, *a, **k):
        self._packed, self._ws = None, {}
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1      # (captured graphs hold the old packs' addresses)
        self.__dict__.pop("_graphs", None)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        # runs for this module on EVERY load, also a recursive one started at a parent (Sam / ProtoSAM.load_state_dict):
        # nn.Module.load_state_dict never calls a child's load_state_dict
        self._packed = None
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self.__dict__.pop("_graphs", None)
        return super()._load_from_state_dict(*a, **k)

    def _pack(self):
        if self._packed is not None:
            return self._packed
        D, oc = self.embed_dim, self.out_chans
        pk = dict(patch_w=f16(self.patch_embed.proj.weight.reshape(D, -1)), patch_b=f32(self.patch_embed.proj.bias),
                  pos=f32(self.pos_embed.reshape(self.grid * self.grid, D)), blocks=[])
        for blk in self.blocks:
            pk["blocks"].append(blk._packed(self.grid))
        pk["neck0"], pk["neck0_lo"] = ops.split_weight_f16(self.neck[0].weight.reshape(oc, D))
        pk["neck2"], pk["neck2_lo"] = ops.split_weight_f16(self.neck[2].weight.permute(0, 2, 3, 1).reshape(oc, 9 * oc))  # [out, (ky,kx,cin)]
        pk["raw"] = {}       # (mean, std) -> patch-embed weights for raw uint8 pixel values (see _patch_raw)
        pk["neck1w"], pk["neck1b"] = f32(self.neck[1].weight), f32(self.neck[1].bias)
        pk["neck3w"], pk["neck3b"] = f32(self.neck[3].weight), f32(self.neck[3].bias)
        self._packed = pk
        return pk

    def _patch_raw(self, pk, mean3, std3):
        """Patch-embedding weights for patches of RAW uint8 pixel values p (exact in fp16): conv((p - mean) / std) = conv'(p) with
        W'[n, c, :] = W[n, c, :] / std[c] (split into an fp16 hi / lo pair) and bias' = bias - sum_c mean[c] * sum(W'[n, c, :])."""
        key = (tuple(float(v) for v in mean3), tuple(float(v) for v in std3))
        if key not in pk["raw"]:
            D = self.embed_dim
            W = self.patch_embed.proj.weight.detach().double().reshape(D, 3, -1)
            mean = torch.tensor(key[0], dtype=torch.float64, device=W.device)
            std = torch.tensor(key[1], dtype=torch.float64, device=W.device)
            Wp = W / std[None, :, None]
            b = self.patch_embed.proj.bias.detach().double() - (Wp.sum(-1) * mean[None, :]).sum(-1)
            hi, lo = ops.split_weight_f16(Wp.reshape(D, -1).float())
            pk["raw"][key] = (hi, lo, b.float().contiguous())
        return pk["raw"][key]

    def _workspace(self, B):
        if B not in self._ws:
            big = max((b for b in self._ws if b > B), default=None)
            if big is not None:   # a sub-batch (e.g. only the slices with a non-empty coarse mask): views of the larger set
                N = self.grid * self.grid
                self._ws[B] = {k: (v[:6 * B * N] if k == "mr" else v[:B] if v.shape[0] == big else v[:B * N]) for k, v in self._ws[big].items()}
                return self._ws[B]
            D, oc, N, H = self.embed_dim, self.out_chans, self.grid * self.grid, self.num_heads
            dev = self.pos_embed.device
            M = B * N
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
            self._ws[B] = dict(x=e((M, D), torch.float32), ln=e((M, D), torch.float16), qkv=e((M, 3 * D), torch.float16),
                               stats=e((M, D // 64, 2), torch.float32), mr=ops.ln_mr_buffer(M, dev),
                               att=e((M, D), torch.float16), hid=e((M, 4 * D), torch.float16),
                               relq=torch.zeros((B, H, N, 2, 32), dtype=torch.float16, device=dev),
                               n0=e((M, oc), torch.float32), n1=e((M, oc), torch.float16),
                               col=e((M, 9 * oc), torch.float16), n2=e((M, oc), torch.float32),
                               out=e((B, N, oc), torch.float32))
        return self._ws[B]

    def _split_buffers(self, ws, M):
        """The lo halves of the split-fp16 neck (`split_fp16`; ~570 MB at 16 slices of ViT-H, col_lo alone 302 MB): allocated when that
        path first runs, like `_rel_buffers`."""
        if "ln_lo" not in ws:
            D, oc, dev = self.embed_dim, self.out_chans, self.pos_embed.device
            ws["ln_lo"] = torch.empty((M, D), dtype=torch.float16, device=dev)
            ws["n1f"] = torch.empty((M, oc), dtype=torch.float32, device=dev)
            ws["n1_lo"] = torch.empty((M, oc), dtype=torch.float16, device=dev)
            ws["col_lo"] = torch.empty((M, 9 * oc), dtype=torch.float16, device=dev)
        return ws

    def _rel_buffers(self, ws, B, H, N):
        """fp32 [B,H,N,64] x 2 of the two-kernel global rel-pos path (268 MB each at 16 slices of ViT-H): only allocated when that path
        runs (the fused kernel does not need them)."""
        if "relh" not in ws:
            dev = self.pos_embed.device
            ws["relh"] = torch.empty((B, H, N, 64), dtype=torch.float32, device=dev)
            ws["relw"] = torch.empty((B, H, N, 64), dtype=torch.float32, device=dev)
        return ws["relh"], ws["relw"]

    def _pack_x3(self, pk):
        """(hi, lo, scale) splits of the blocks' four Linear weights for ops.gemm_f32x3 (one-time, only when `gemm_x3` runs)."""
        if "x3" not in pk:
            def sp(w):
                sc = ops.split_scale_for(w, ops.X3_WEIGHT_SCALE)
                return ops.split_weight_f16(w, sc) + (sc,)
            pk["x3"] = [dict(qkv=sp(b.attn.qkv.weight), proj=sp(b.attn.proj.weight), l1=sp(b.mlp.lin1.weight), l2=sp(b.mlp.lin2.weight))
                        for b in self.blocks]
        return pk["x3"]

    def _encode_blocks_x3(self, pk, ws, B):
        """The block stack with every Linear at fp32 accuracy (see `gemm_x3`): x fp32 [M, D] in ws["x"], updated in place."""
        D, H, N, g = self.embed_dim, self.num_heads, self.grid * self.grid, self.grid
        M = B * N
        x = ws["x"]
        if "ln32" not in ws:
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=x.device)  # noqa: E731
            ws["ln32"], ws["qkv32"], ws["hid32"] = e((M, D), torch.float32), e((M, 3 * D), torch.float32), e((M, 4 * D), torch.float32)
        ln32, qkv32, hid32 = ws["ln32"], ws["qkv32"], ws["hid32"]
        for blk, bp, xp in zip(self.blocks, pk["blocks"], self._pack_x3(pk)):
            ops.layernorm(x, bp["n1w"], bp["n1b"], LN_EPS, out=ln32, out_dtype=torch.float32)
            ops.gemm_f32x3(ln32, xp["qkv"], bp["qkv_b"], out=qkv32)
            ops.cast_f16(qkv32, ws["qkv"])                                          # the attention kernels' operand format
            blk.attn._attend(ws["qkv"], B, g, bp["ws"], bp["rpack"], bp["pad_row"], out=ws["att"], relq=ws["relq"],
                             rel_bufs=lambda: self._rel_buffers(ws, B, H, N))
            ops.cast_f32(ws["att"], ln32)
            ops.gemm_f32x3(ln32, xp["proj"], bp["proj_b"], out=x, resid=x)
            ops.layernorm(x, bp["n2w"], bp["n2b"], LN_EPS, out=ln32, out_dtype=torch.float32)
            ops.gemm_f32x3(ln32, xp["l1"], bp["l1b"], out=hid32)
            ops.gelu_f32_(hid32)
            ops.gemm_f32x3(hid32, xp["l2"], bp["l2b"], out=x, resid=x)

    def encode_patches(self, patches, B, raw_norm=None):
        """One or two slices: a captured HIP graph of the forward's launches (ops.GraphCache; ~330 for ViT-H), else `_encode_patches`.
        raw_norm = (mean3, std3): `patches` hold RAW uint8 pixel values (exact in fp16) and the normalisation is folded into the
        patch-embedding weights (`split_fp16`)."""
        if B <= 2 and not self.gemm_x3 and ops.graph_wanted(patches, 2 * self.grid * self.grid):
            gc = self.__dict__.setdefault("_graphs", ops.GraphCache("the SAM image encoder forward"))
            key = (tuple(patches.shape), B, str(patches.device), getattr(self, "_weights_epoch", 0), self.fold_ln,
                   getattr(self, "fold_min_fill", None), ops.dispatch_key(), self.split_fp16, self.splitk_lin2,
                   None if raw_norm is None else (tuple(raw_norm[0]), tuple(raw_norm[1])))
            out = gc.run(key, patches, lambda t: self._encode_patches(t, B, raw_norm))
            if out is not None:
                return out
        return self._encode_patches(patches, B, raw_norm)

    def _encode_patches(self, patches, B, raw_norm=None):
        """patches fp16 [B*4096, 3*16*16] (im2col of the normalised image) -> token-major embedding fp32 [B,4096,out].
        `fold_ln` (optional, see __init__): the blocks' LayerNorms never run as passes of their own - the GEMM that updates the residual
        stream also emits half(x) and per-row partial sums, `ln_finalize` turns them into (mean, rstd), and the consuming
        GEMM (qkv / lin1, weights pre-multiplied by the LayerNorm weight) applies them in its epilogue. Same arithmetic up to
        rounding (tools/emulate_ln_fusion.py: embedding error 6.4e-4 mean vs 7.0e-4 for the separate LayerNorm pass)."""
        pk = self._pack()
        ws = self._workspace(B)
        D, H, N, g = self.embed_dim, self.num_heads, self.grid * self.grid, self.grid
        hd = D // H
        x = ws["x"]
        x3 = self.gemm_x3 and not ops.QKV_HEAD_MAJOR
        fold = self.fold_ln and not ops.QKV_HEAD_MAJOR and D % 64 == 0 and not x3
        M = B * N
        fold = fold and ops.fold_pays(M, D, x.device, self.fold_min_fill)   # (small launches: separate passes are faster)
        x16, stats, mr = ws["ln"], ws["stats"], ws["mr"]
        fk = dict(out16=x16, stats=stats) if fold else {}
        if raw_norm is not None:     # exact uint8 pixel values against the split (hi + lo) weights of the folded normalisation
            pw_hi, pw_lo, pb = self._patch_raw(pk, *raw_norm)
            if "patch" in self._split_parts:
                ops.gemm(patches, pw_hi, pb, out=x, epilogue=ops.EPI_F32, resid=pk["pos"], resid_mod=N)
                ops.gemm(patches, pw_lo, None, out=x, epilogue=ops.EPI_F32, resid=x, **fk)
            else:
                ops.gemm(patches, pw_hi, pb, out=x, epilogue=ops.EPI_F32, resid=pk["pos"], resid_mod=N, **fk)
        else:
            ops.gemm(patches, pk["patch_w"], pk["patch_b"], out=x, epilogue=ops.EPI_F32, resid=pk["pos"], resid_mod=N, **fk)
        if x3:
            self._encode_blocks_x3(pk, ws, B)
        # split-K form of mlp.lin2 (see __init__): only without the fold (one or two images), only where the library says it pays
        sk = ops.gemm_splitk_ranges(M, D, 4 * D) if (self.splitk_lin2 and not fold and not x3 and not ops.QKV_HEAD_MAJOR) else 0
        if sk >= 2 and "sk_ws" not in ws:
            ws["sk_ws"] = torch.empty(sk * ops.splitk_rows(M) * D, dtype=torch.float32, device=x.device)
        ln1_ready = False                                   # ws["ln"] already holds norm1(x) of the block about to run
        nblk = len(pk["blocks"])
        for bi, (blk, bp) in enumerate(zip(self.blocks if not x3 else (), pk["blocks"])):
            if fold:
                ops.ln_finalize(stats, M, D, LN_EPS, mr=mr)
                ops.gemm(x16, bp["qkv_wf"], bp["qkv_t"], out=ws["qkv"], epilogue=ops.EPI_F16, ln_mr=mr, ln_s=bp["qkv_s"])
            elif ops.QKV_HEAD_MAJOR:
                ops.layernorm(x, bp["n1w"], bp["n1b"], LN_EPS, out=ws["ln"])
                ops.gemm_heads(ws["ln"], bp["qkv_w"], bp["qkv_b"], hd, out=ws["qkv"])   # [3,H,B*N,hd]
            else:
                if not ln1_ready:
                    ops.layernorm(x, bp["n1w"], bp["n1b"], LN_EPS, out=ws["ln"])
                ops.gemm(ws["ln"], bp["qkv_w"], bp["qkv_b"], out=ws["qkv"], epilogue=ops.EPI_F16)
            blk.attn._attend(ws["qkv"], B, g, bp["ws"], bp["rpack"], bp["pad_row"], out=ws["att"], relq=ws["relq"],
                             rel_bufs=lambda: self._rel_buffers(ws, B, H, N))
            ops.gemm(ws["att"], bp["proj_w"], bp["proj_b"], out=x, epilogue=ops.EPI_F32, resid=x, **fk)
            if fold:
                ops.ln_finalize(stats, M, D, LN_EPS, mr=mr)
                ops.gemm(x16, bp["l1wf"], bp["l1t"], out=ws["hid"], epilogue=ops.EPI_GELU_F16, ln_mr=mr, ln_s=bp["l1s"])
            else:
                ops.layernorm(x, bp["n2w"], bp["n2b"], LN_EPS, out=ws["ln"])
                ops.gemm(ws["ln"], bp["l1w"], bp["l1b"], out=ws["hid"], epilogue=ops.EPI_GELU_F16)
            if sk >= 2:      # x += lin2(hid); ws["ln"] = the next block's norm1(x), or fp16(x) for the neck behind the last block
                nxt = pk["blocks"][bi + 1] if bi + 1 < nblk else None
                ops.gemm_splitk_ln(ws["hid"], bp["l2w"], bp["l2b"], x, sk, ws["sk_ws"], nxt["n1w"] if nxt else None,
                                   nxt["n1b"] if nxt else None, LN_EPS, out16=ws["ln"])
                ln1_ready = True
            else:
                ops.gemm(ws["hid"], bp["l2w"], bp["l2b"], out=x, epilogue=ops.EPI_F32, resid=x, **fk)
        neck_cast_done = sk >= 2 and nblk > 0 and not x3
        # neck (image_encoder.py:90-106): the residual stream goes through the 1x1-conv GEMM as fp16 (with `fold_ln` the last
        # lin2 epilogue already wrote that copy)
        xh = ws["ln"]
        if (self.split_fp16 and "neck" in self._split_parts) or x3:
            # (hi, lo) pairs of the activations and of the weights: hi W_hi + lo W_hi + hi W_lo, accumulated in the fp32 output
            self._split_buffers(ws, M)
            ops.split_f16(x, hi=xh, lo=ws["ln_lo"], write_hi=not fold)
            ops.gemm(xh, pk["neck0"], None, out=ws["n0"], epilogue=ops.EPI_F32)
            ops.gemm(ws["ln_lo"], pk["neck0"], None, out=ws["n0"], epilogue=ops.EPI_F32, resid=ws["n0"])
            ops.gemm(xh, pk["neck0_lo"], None, out=ws["n0"], epilogue=ops.EPI_F32, resid=ws["n0"])
            ops.layernorm(ws["n0"], pk["neck1w"], pk["neck1b"], LN_EPS, out=ws["n1"], out2=ws["n1f"])
            ops.split_f16(ws["n1f"], hi=ws["n1"], lo=ws["n1_lo"], write_hi=False)
            ops.im2col3x3(ws["n1"], B, g, g, self.out_chans, out=ws["col"])
            ops.im2col3x3(ws["n1_lo"], B, g, g, self.out_chans, out=ws["col_lo"])
            ops.gemm(ws["col"], pk["neck2"], None, out=ws["n2"], epilogue=ops.EPI_F32)
            ops.gemm(ws["col_lo"], pk["neck2"], None, out=ws["n2"], epilogue=ops.EPI_F32, resid=ws["n2"])
            ops.gemm(ws["col"], pk["neck2_lo"], None, out=ws["n2"], epilogue=ops.EPI_F32, resid=ws["n2"])
        else:
            if not fold and not neck_cast_done:
                ops.cast_f16(x, xh)
            ops.gemm(xh, pk["neck0"], None, out=ws["n0"], epilogue=ops.EPI_F32)
            ops.layernorm(ws["n0"], pk["neck1w"], pk["neck1b"], LN_EPS, out=ws["n1"])
            ops.im2col3x3(ws["n1"], B, g, g, self.out_chans, out=ws["col"])
            ops.gemm(ws["col"], pk["neck2"], None, out=ws["n2"], epilogue=ops.EPI_F32)
        ops.layernorm(ws["n2"], pk["neck3w"], pk["neck3b"], LN_EPS, out=ws["out"].view(B * N, self.out_chans),
                      out_dtype=torch.float32)
        return ws["out"]

    def forward_tokens(self, x):
        """x fp32 [B,3,1024,1024] (already normalised / padded) -> token-major [B, 4096, out_chans]."""
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (3, self.img_size, self.img_size)
        P = self.patch_size
        patches = ops.patchify_bilinear(x.float().contiguous(), self.img_size, P, 3 * P * P)
        return self.encode_patches(patches, B)

    def forward(self, x):
        t = self.forward_tokens(x)
        B = t.shape[0]
        return t.view(B, self.grid, self.grid, self.out_chans).permute(0, 3, 1, 2)
