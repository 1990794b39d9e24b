"""DINOv2 ViT encoder on the HIP kernels, with the hub model's state-dict names and call contract.

The reference obtains this model with `torch.hub.load('facebookresearch/dinov2', 'dinov2_vitb14')`
(/root/reference/models/grid_proto_fewshot.py:54-72) and only ever calls
`encoder.forward_features(x)["x_norm_patchtokens"]` (grid_proto_fewshot.py:90-91). This class keeps that
contract and the hub parameter names (`cls_token, pos_embed, mask_token, patch_embed.proj.*,
blocks.{i}.{norm1,norm2}.*, blocks.{i}.attn.{qkv,proj}.*, blocks.{i}.ls{1,2}.gamma,
blocks.{i}.mlp.fc{1,2}.*, norm.*`, `register_tokens` for the `_reg` variant) so a real checkpoint loads with
`load_state_dict(strict=True)`.

Execution (all arithmetic in csrc/*.hip): bilinear-resize+im2col -> patch GEMM (+bias +pos-embed, rows
remapped behind the cls token) -> per block [LN -> QKV GEMM -> fused attention -> proj GEMM with
LayerScale+residual epilogue -> LN -> fc1 GEMM with GELU epilogue -> fc2 GEMM with LayerScale+residual
epilogue] -> final LN (fp32). GEMM/attention operands are fp16 with fp32 accumulation; the residual stream
and all LayerNorm statistics are fp32. The nn.Module tree below only holds parameters.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

DINO_CFGS = {
    "dinov2_vitb14": dict(embed_dim=768, depth=12, num_heads=12, num_register_tokens=0, interpolate_antialias=False,
                          interpolate_offset=0.1),
    "dinov2_vitl14": dict(embed_dim=1024, depth=24, num_heads=16, num_register_tokens=0, interpolate_antialias=False,
                          interpolate_offset=0.1),
    "dinov2_vitl14_reg": dict(embed_dim=1024, depth=24, num_heads=16, num_register_tokens=4,
                              interpolate_antialias=True, interpolate_offset=0.0),
}
PATCH = 14
LN_EPS = 1e-6


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.attn = _Attention(dim)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = _Mlp(dim, dim * 4)
        self.ls2 = _LayerScale(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=PATCH, stride=PATCH)


class DinoVisionTransformer(nn.Module):
    def __init__(self, name="dinov2_vitb14", depth=None):
        super().__init__()
        cfg = dict(DINO_CFGS[name])
        if depth is not None:
            cfg["depth"] = depth
        self.cfg = cfg
        D = cfg["embed_dim"]
        self.embed_dim = D
        self.num_heads = cfg["num_heads"]
        self.patch_size = PATCH
        self.num_register_tokens = cfg["num_register_tokens"]
        self.patch_embed = _PatchEmbed(D)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + 37 * 37, D))  # img_size 518 / 14
        self.mask_token = nn.Parameter(torch.zeros(1, D))              # unused at inference, kept for strict loading
        if self.num_register_tokens:
            self.register_tokens = nn.Parameter(torch.zeros(1, self.num_register_tokens, D))
        self.blocks = nn.ModuleList([_Block(D) for _ in range(cfg["depth"])])
        self.norm = nn.LayerNorm(D, eps=LN_EPS)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        self._packed = None
        self._pos_cache = {}
        self._ws = {}
        # LayerNorm folded into the GEMMs either side of it (ops.gemm ... out16 / stats / ln_mr / ln_s): on by default since the
        # assembly GEMM has the folded epilogues (round 3: 145.3 vs 142.8 slices/s, same box, interleaved; the LayerNorm passes of a
        # 16-slice step cost 6 ms, the fp16 copy + row sums in the producers' epilogues and the rank-1 correction MFMAs in the
        # consumers' 3.8 ms; it also lowers the embedding error). Round 2 measured the opposite on the HIP kernels (122.2 vs 124.3:
        # +10 ms of epilogue). PSAM_FOLD_LN=0 / `fold_ln = False` selects the separate passes.
        self.fold_ln = os.environ.get("PSAM_FOLD_LN", "1") != "0"
        self.fold_min_fill = float(os.environ.get("PSAM_FOLD_MIN_FILL", "0.8"))      # ... where the launches fill the CUs (ops.fold_pays); 0 = always

    # -- weight packing (fp16 GEMM operands); rebuilt whenever parameters change -----------------------------
    def _apply(self, fn, *a, **k):
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self._packed = None
        self._pos_cache = {}
        self._ws = {}
        self.__dict__.pop("_graphs", None)          # (captured graphs hold the old packs' / workspaces' addresses)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load (FewShotSeg.load_state_dict)
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self._packed = None
        self._pos_cache = {}
        self.__dict__.pop("_graphs", None)
        return super()._load_from_state_dict(*a, **k)

    def _pack(self):
        if self._packed is not None:
            return self._packed
        D = self.embed_dim
        K = 3 * PATCH * PATCH
        Kpad = (K + 63) // 64 * 64
        w = torch.zeros((D, Kpad), dtype=torch.float16, device=self.pos_embed.device)
        w[:, :K] = self.patch_embed.proj.weight.detach().reshape(D, K).half()
        pk = dict(Kpad=Kpad, patch_w=w, patch_b=self.patch_embed.proj.bias.detach().float().contiguous(), blocks=[])
        for blk in self.blocks:
            pk["blocks"].append(dict(
                qkv_w=blk.attn.qkv.weight.detach().half().contiguous(), qkv_b=blk.attn.qkv.bias.detach().float().contiguous(),
                proj_w=blk.attn.proj.weight.detach().half().contiguous(), proj_b=blk.attn.proj.bias.detach().float().contiguous(),
                fc1_w=blk.mlp.fc1.weight.detach().half().contiguous(), fc1_b=blk.mlp.fc1.bias.detach().float().contiguous(),
                fc2_w=blk.mlp.fc2.weight.detach().half().contiguous(), fc2_b=blk.mlp.fc2.bias.detach().float().contiguous(),
                n1w=blk.norm1.weight.detach().float().contiguous(), n1b=blk.norm1.bias.detach().float().contiguous(),
                n2w=blk.norm2.weight.detach().float().contiguous(), n2b=blk.norm2.bias.detach().float().contiguous(),
                g1=blk.ls1.gamma.detach().float().contiguous(), g2=blk.ls2.gamma.detach().float().contiguous()))
            # LayerNorm folded into the consuming GEMM (ops.fold_layernorm)
            d = pk["blocks"][-1]
            d["qkv_wf"], d["qkv_s"], d["qkv_t"] = ops.fold_layernorm(blk.attn.qkv.weight, blk.attn.qkv.bias, blk.norm1.weight,
                                                                    blk.norm1.bias)
            d["fc1_wf"], d["fc1_s"], d["fc1_t"] = ops.fold_layernorm(blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.norm2.weight,
                                                                    blk.norm2.bias)
        pk["nw"] = self.norm.weight.detach().float().contiguous()
        pk["nb"] = self.norm.bias.detach().float().contiguous()
        self._packed = pk
        return pk

    def _pos_for_grid(self, g):
        """Input-size-only, computed once per grid size at setup (host-side PyTorch, cached): bicubic resample of
        the 37x37 pos-embed with the hub's scale_factor=(g+offset)/37 rule. Returns (pos_patches [g*g, D],
        prefix rows [1+R, D] = cls+pos[0] followed by the register tokens)."""
        if g in self._pos_cache:
            return self._pos_cache[g]
        pe = self.pos_embed.detach().float()
        N = pe.shape[1] - 1
        M = int(math.sqrt(N))
        D = pe.shape[-1]
        if g * g == N:
            grid = pe[0, 1:]
        else:
            off = self.cfg["interpolate_offset"]
            t = pe[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)
            if off:
                kw = dict(scale_factor=(float(g + off) / M, float(g + off) / M))
            else:
                kw = dict(size=(g, g))
            t = F.interpolate(t.cpu(), mode="bicubic", antialias=self.cfg["interpolate_antialias"], **kw)
            assert tuple(t.shape[-2:]) == (g, g)
            grid = t.permute(0, 2, 3, 1).reshape(g * g, D).to(pe.device)
        prefix = (self.cls_token.detach().float()[0] + pe[0, :1])
        if self.num_register_tokens:
            prefix = torch.cat([prefix, self.register_tokens.detach().float()[0]], dim=0)
        out = (grid.contiguous(), prefix.contiguous())
        self._pos_cache[g] = out
        return out

    def _workspace(self, B, N):
        key = (B, N)
        if key not in self._ws:
            D, dev = self.embed_dim, self.pos_embed.device
            M = B * N
            self._ws[key] = dict(
                x=torch.empty((B, N, D), dtype=torch.float32, device=dev),
                ln=torch.empty((M, D), dtype=torch.float16, device=dev),
                stats=torch.empty((M, D // 64, 2), dtype=torch.float32, device=dev),
                mr=ops.ln_mr_buffer(M, dev),
                qkv=torch.empty((M, 3 * D), dtype=torch.float16, device=dev),
                att=torch.empty((M, D), dtype=torch.float16, device=dev),
                hid=torch.empty((M, 4 * D), dtype=torch.float16, device=dev),
                out=torch.empty((B, N, D), dtype=torch.float32, device=dev))
        return self._ws[key]

    # -- forward ---------------------------------------------------------------------------------------------------
    def forward_tokens(self, imgs, S):
        """imgs fp32 [B,3,H,W] (any H,W) -> bilinear to SxS -> final-norm tokens fp32 [B, 1+R+n, D] (workspace).
        Small batches replay a captured HIP graph of the ~100 launches (`_graph_tokens`): with one or two slices per call the
        kernels are shorter than the ~7 us a launch costs the host, and the GPU idled 42 % of such a forward (ops.GraphCache)."""
        if ops.graph_wanted(imgs, 2):
            gc = self.__dict__.setdefault("_graphs", ops.GraphCache("the DINOv2 forward"))
            key = (tuple(imgs.shape), S, str(imgs.device), getattr(self, "_weights_epoch", 0), self.fold_ln, self.fold_min_fill,
                   ops.dispatch_key())
            out = gc.run(key, imgs.float(), lambda t: self._forward_tokens(t, S))
            if out is not None:
                return out
        return self._forward_tokens(imgs, S)

    def _forward_tokens(self, imgs, S):
        assert S % PATCH == 0
        pk = self._pack()
        B = imgs.shape[0]
        g = S // PATCH
        n = g * g
        R = self.num_register_tokens
        N = 1 + R + n
        D, H = self.embed_dim, self.num_heads
        hd = D // H
        ws = self._workspace(B, N)
        x = ws["x"]
        pos, prefix = self._pos_for_grid(g)
        patches = ops.patchify_bilinear(imgs.contiguous(), S, PATCH, pk["Kpad"])
        ops.gemm(patches, pk["patch_w"], pk["patch_b"], out=x.view(B * N, D), epilogue=ops.EPI_F32, resid=pos,
                 resid_mod=n, out_seg=n, out_seg_stride=N, out_seg_off=1 + R)
        for r in range(1 + R):
            ops.broadcast_rows(prefix[r], x, B, N * D, r * D)
        x2 = x.view(B * N, D)
        # `fold_ln`: as in the SAM encoder (image_encoder.py), the residual-stream GEMMs emit half(x) + row partial sums and the
        # consuming GEMM applies (mean, rstd) in its epilogue. The first norm1 stays a pass of its own: the cls / register rows
        # come from `broadcast_rows`, not from a GEMM epilogue.
        fold = self.fold_ln and not ops.QKV_HEAD_MAJOR and D % 64 == 0
        M = B * N
        fold = fold and ops.fold_pays(M, D, x.device, self.fold_min_fill)   # (small launches: separate passes are faster)
        M = B * N
        x16, stats, mr = ws["ln"], ws["stats"], ws["mr"]
        fk = dict(out16=x16, stats=stats) if fold else {}
        for bi, bp in enumerate(pk["blocks"]):
            if fold and bi > 0:
                ops.ln_finalize(stats, M, D, LN_EPS, mr=mr)
                ops.gemm(x16, bp["qkv_wf"], bp["qkv_t"], out=ws["qkv"], epilogue=ops.EPI_F16, ln_mr=mr, ln_s=bp["qkv_s"])
            else:
                ops.layernorm(x2, bp["n1w"], bp["n1b"], LN_EPS, out=ws["ln"])
                if ops.QKV_HEAD_MAJOR:
                    ops.gemm_heads(ws["ln"], bp["qkv_w"], bp["qkv_b"], hd, out=ws["qkv"])   # [3,H,B*N,hd]
                else:
                    ops.gemm(ws["ln"], bp["qkv_w"], bp["qkv_b"], out=ws["qkv"], epilogue=ops.EPI_F16)
            ops.attention(ws["qkv"], B, N, H, hd, hd ** -0.5, out=ws["att"], head_major=ops.QKV_HEAD_MAJOR)
            ops.gemm(ws["att"], bp["proj_w"], bp["proj_b"], out=x2, epilogue=ops.EPI_F32, resid=x2, gamma=bp["g1"], **fk)
            if fold:
                ops.ln_finalize(stats, M, D, LN_EPS, mr=mr)
                ops.gemm(x16, bp["fc1_wf"], bp["fc1_t"], out=ws["hid"], epilogue=ops.EPI_GELU_F16, ln_mr=mr, ln_s=bp["fc1_s"])
            else:
                ops.layernorm(x2, bp["n2w"], bp["n2b"], LN_EPS, out=ws["ln"])
                ops.gemm(ws["ln"], bp["fc1_w"], bp["fc1_b"], out=ws["hid"], epilogue=ops.EPI_GELU_F16)
            ops.gemm(ws["hid"], bp["fc2_w"], bp["fc2_b"], out=x2, epilogue=ops.EPI_F32, resid=x2, gamma=bp["g2"], **fk)
        ops.layernorm(x2, pk["nw"], pk["nb"], LN_EPS, out=ws["out"].view(B * N, D), out_dtype=torch.float32)
        return ws["out"]

    def forward_features(self, x):
        """Hub contract: x [B,3,H,W] with H, W multiples of 14 -> dict with x_norm_clstoken / x_norm_patchtokens."""
        B, _, Hh, Ww = x.shape
        assert Hh == Ww and Hh % PATCH == 0, "square inputs with side a multiple of 14"
        t = self.forward_tokens(x.float(), Hh)
        R = self.num_register_tokens
        return {"x_norm_clstoken": t[:, 0], "x_norm_regtokens": t[:, 1:1 + R], "x_norm_patchtokens": t[:, 1 + R:]}

    def forward(self, x):
        return self.forward_features(x)["x_norm_clstoken"]
