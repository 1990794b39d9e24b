"""SAM's two-way transformer on the HIP kernels (names of models/segment_anything/modeling/transformer.py:16-240).

`TwoWayTransformer.run_tokens` is the decoder's hot path (MaskDecoder.predict_masks_tokens drives it): B prompt sets against
the token-major embedding(s) of their image(s), everything in buffers of a per-shape workspace. Token side (T <= 16 tokens
per prompt set): fp32 `small_linear` / `small_attention`. Image side (4096 tokens): the k / v / q projections and the
image-to-token out-projection are GEMMs at fp32 accuracy (`psam_gemm_f32x3`, or the exact-fp32 MFMA `psam_gemm_f32`; `keys + pe`
is added on the operand load); token-to-image attention is `psam_t2i_attention`.

The modules' own `forward`s (transformer.py:62-106 TwoWayTransformer, :151-182 TwoWayAttentionBlock, :218-240 Attention) are
thin drivers of the same kernels with the reference's signatures and return values, for callers that use the containers
directly; the hot path never goes through them.
"""
import torch
import torch.nn as nn

from ... import ops
from .common import MLPBlock, f16, f32

LN_EPS = 1e-5  # nn.LayerNorm default (transformer.py:133-144)


class Attention(nn.Module):
    def __init__(self, embedding_dim, num_heads, downsample_rate=1):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)
        self._cache = {}

    def _apply(self, fn, *a, **k):
        self._cache = {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load
        self._cache = {}
        return super()._load_from_state_dict(*a, **k)

    def _pack(self, image_side=()):
        """fp32 weights; for the projections named in `image_side` (applied to the 4096 image tokens) also the fp16 copy and the
        (hi, lo, scale) split of ops.gemm_f32x3."""
        key = tuple(image_side)
        if key not in self._cache:
            d = dict(qw=f32(self.q_proj.weight), qb=f32(self.q_proj.bias), kw=f32(self.k_proj.weight), kb=f32(self.k_proj.bias),
                     vw=f32(self.v_proj.weight), vb=f32(self.v_proj.bias), ow=f32(self.out_proj.weight), ob=f32(self.out_proj.bias))
            mods = {"qw": self.q_proj, "kw": self.k_proj, "vw": self.v_proj, "ow": self.out_proj}
            for nm in image_side:
                d[nm + "16"] = f16(mods[nm].weight)
                sc = ops.split_scale_for(mods[nm].weight, ops.X3_WEIGHT_SCALE)       # (2^8 unless a weight would leave the fp16 range)
                d[nm + "x3"] = ops.split_weight_f16(mods[nm].weight, sc) + (sc,)      # (hi, lo, scale) for ops.gemm_f32x3
            self._cache[key] = d
        return self._cache[key]

    @torch.no_grad()
    def forward(self, q, k, v):
        """transformer.py:218-240: out_proj(softmax(q_proj(q) k_proj(k)^T / sqrt(c)) v_proj(v)); q [B,Nq,C], k / v [B,Nk,C] fp32 ->
        [B,Nq,C]. Kernels by shape: <= 16 keys `psam_small_attention` (token self-attention, image-to-token), otherwise <= 16
        queries over <= 4096 keys `psam_t2i_attention` (token-to-image, 16-wide heads)."""
        ap = self._pack()
        B, Nq, C = q.shape
        Nk = k.shape[1]
        NH, ci = self.num_heads, self.internal_dim
        q2, k2, v2 = (t.reshape(-1, C).float().contiguous() for t in (q, k, v))
        qp = ops.small_linear(q2, ap["qw"], ap["qb"])
        kp = ops.small_linear(k2, ap["kw"], ap["kb"])
        vp = ops.small_linear(v2, ap["vw"], ap["vb"])
        att = torch.empty_like(qp)
        if Nk <= 16:
            ops.small_attention(qp, kp, vp, att, B, Nq, Nk, NH, ci // NH, ci, ci, ci, ci)
        elif Nq <= 16 and Nk <= 4096 and ci // NH == 16:
            ops.t2i_attention(qp, kp, vp, att, B, Nq, Nk, NH)
        else:
            raise NotImplementedError(f"Attention.forward: {Nq} queries x {Nk} keys with {ci // NH}-wide heads has no HIP kernel "
                                      "(<= 16 keys, or <= 16 queries over <= 4096 keys with 16-wide heads)")
        return ops.small_linear(att, ap["ow"], ap["ob"]).view(B, Nq, C)


class _Run:
    """Buffers and switches of one `run_tokens` call, shared by the layers."""

    def __init__(self, ws, B, T, Nk, NH, tok_pe, pe_tok, h16, x3):
        self.ws, self.B, self.T, self.Nk, self.NH, self.tok_pe, self.pe_tok, self.h16, self.x3 = ws, B, T, Nk, NH, tok_pe, pe_tok, h16, x3
        # K / V of the token-to-image attention head-major (contiguous 64-byte key rows; the kernels are written for 16-wide heads)
        self.hm = (not h16) and T <= 16 and Nk >= 64 and 128 // NH == 16
        self.t2i_s = ops.t2i_split(B, NH, T, Nk) if NH == 8 else 1     # key ranges per (prompt set, head): few prompt sets leave the CUs idle

    def img_proj(self, w_name, ap, out, with_pe, heads=None):
        """image-token projection: (keys [+ key_pe]) @ W^T + b (transformer.py:228-230 of the 4096-token operand); heads = (Nk, hd):
        written head-major [B][NH][Nk][hd] for the token-to-image attention kernel"""
        ws = self.ws
        if self.h16:
            ops.gemm(ws["kpe16"] if with_pe else ws["k16"], ap[w_name + "16"], ap[w_name[0] + "b"], out=out, epilogue=ops.EPI_F16)
        elif self.x3:
            ops.gemm_f32x3(ws["keys"], ap[w_name + "x3"], ap[w_name[0] + "b"], out=out, a2=self.pe_tok if with_pe else None,
                           a2_mod=self.Nk, heads=heads)
        else:
            ops.gemm_f32(ws["keys"], ap[w_name], ap[w_name[0] + "b"], out=out, a2=self.pe_tok if with_pe else None, a2_mod=self.Nk,
                         heads=heads)

    def t2i(self, ap, resid_ln):
        """queries += attention(q = queries + query_pe, k = keys + key_pe, v = keys), then LayerNorm (transformer.py:163-167,98-104)"""
        ws, B, T, Nk, NH = self.ws, self.B, self.T, self.Nk, self.NH
        q = ws["q"]
        heads = (Nk, 128 // NH) if self.hm else None
        ops.small_linear(q, ap["qw"], ap["qb"], out=ws["tq128"], x2=self.tok_pe)
        self.img_proj("kw", ap, ws["p0"], True, heads=heads)
        self.img_proj("vw", ap, ws["p1"], False, heads=heads)
        ops.t2i_attention(ws["tq128"], ws["p0"], ws["p1"], ws["ta128"], B, T, Nk, NH, head_major=self.hm,
                          split=(self.t2i_s, ws["t2i_part"]))
        ops.small_linear(ws["ta128"], ap["ow"], ap["ob"], out=ws["t1"], resid=q)
        ops.layernorm(ws["t1"], resid_ln[0], resid_ln[1], LN_EPS, out=q, out_dtype=torch.float32)


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, embedding_dim, num_heads, mlp_dim=2048, activation=nn.ReLU, attention_downsample_rate=2,
                 skip_first_layer_pe=False):
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe
        self.num_heads = num_heads
        self._cache = None
        self._ws = {}

    def _apply(self, fn, *a, **k):
        self._cache, self._ws = None, {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._cache = None
        return super()._load_from_state_dict(*a, **k)

    def _packed(self):
        if self._cache is None:
            if not isinstance(self.mlp.act, nn.ReLU):
                raise NotImplementedError("the two-way block's MLP runs with ReLU (build_sam.py: TwoWayTransformer's default)")
            self._cache = dict(
                sa=self.self_attn._pack(), t2i=self.cross_attn_token_to_image._pack(("kw", "vw")),
                i2t=self.cross_attn_image_to_token._pack(("qw", "ow")),
                n=[(f32(n.weight), f32(n.bias)) for n in (self.norm1, self.norm2, self.norm3, self.norm4)],
                l1w=f32(self.mlp.lin1.weight), l1b=f32(self.mlp.lin1.bias), l2w=f32(self.mlp.lin2.weight),
                l2b=f32(self.mlp.lin2.bias), skip_pe=self.skip_first_layer_pe)
        return self._cache

    def _run(self, r, src):
        """transformer.py:151-182 on the buffers of `r`: src = the block's input queries (fp32 [B*T,256]; the result goes to
        ws["q"]), keys in ws["keys"] (+ the fp16 copies of the fp16 image side)."""
        L = self._packed()
        ws, B, T, Nk, NH = r.ws, r.B, r.T, r.Nk, r.NH
        q, lin = ws["q"], ops.small_linear
        sa = L["sa"]
        x2 = None if L["skip_pe"] else r.tok_pe
        lin(src, sa["qw"], sa["qb"], out=ws["tq"], x2=x2)
        lin(src, sa["kw"], sa["kb"], out=ws["tk"], x2=x2)
        lin(src, sa["vw"], sa["vb"], out=ws["tv"])
        ops.small_attention(ws["tq"], ws["tk"], ws["tv"], ws["ta"], B, T, T, NH, 256 // NH, 256, 256, 256, 256)
        lin(ws["ta"], sa["ow"], sa["ob"], out=ws["t1"], resid=None if L["skip_pe"] else src)
        ops.layernorm(ws["t1"], L["n"][0][0], L["n"][0][1], LN_EPS, out=q, out_dtype=torch.float32)
        r.t2i(L["t2i"], L["n"][1])
        lin(q, L["l1w"], L["l1b"], out=ws["hid"], act=1)
        if B * T >= 32:      # 2048 -> 256 on few rows: eight K ranges side by side (psam_small_linear_splitk)
            ops.small_linear_splitk(ws["hid"], L["l2w"], L["l2b"], q, ws["t1"], ws["parts"], 8)
        else:
            lin(ws["hid"], L["l2w"], L["l2b"], out=ws["t1"], resid=q)
        ops.layernorm(ws["t1"], L["n"][2][0], L["n"][2][1], LN_EPS, out=q, out_dtype=torch.float32)
        ia = L["i2t"]
        keys = ws["keys"]
        r.img_proj("qw", ia, ws["p0"], True)
        lin(q, ia["kw"], ia["kb"], out=ws["tk"][:, :128], x2=r.tok_pe)
        lin(q, ia["vw"], ia["vb"], out=ws["tv"][:, :128])
        ops.small_attention(ws["p0"], ws["tk"][:, :128], ws["tv"][:, :128], ws["p1"], B, Nk, T, NH, 128 // NH, 128,
                            256, 256, 128)
        if r.h16:
            ops.gemm(ws["p1"], ia["ow16"], ia["ob"], out=keys, epilogue=ops.EPI_F32, resid=keys)
        elif r.x3:
            ops.gemm_f32x3(ws["p1"], ia["owx3"], ia["ob"], out=keys, resid=keys)
        else:
            ops.gemm_f32(ws["p1"], ia["ow"], ia["ob"], out=keys, resid=keys)
        ops.ln_pe(keys, r.pe_tok, B * Nk, y32=keys, y16=ws["k16"], ype16=ws["kpe16"], w=L["n"][3][0], b=L["n"][3][1], pe_mod=Nk,
                  eps=LN_EPS)

    @torch.no_grad()
    def forward(self, queries, keys, query_pe, key_pe):
        """transformer.py:151-182: (queries [B,T,C], keys [B,Nk,C], query_pe [B,T,C], key_pe [B,Nk,C] - one positional grid shared
        by the batch, as SAM passes it) -> (queries, keys)."""
        B, T, C = queries.shape
        Nk = keys.shape[1]
        _check_shapes(C, T, Nk, self.num_heads)
        dev = queries.device
        ws = _workspace(self._ws, B, T, Nk, dev, False)
        pe_tok = _shared_pe(key_pe, Nk)
        tok_pe = query_pe.reshape(B * T, C).float().contiguous()
        r = _Run(ws, B, T, Nk, self.num_heads, tok_pe, pe_tok, False, True)
        ws["keys"].copy_(keys.reshape(B * Nk, C))
        self._run(r, queries.reshape(B * T, C).float().contiguous())
        return ws["q"].view(B, T, C).clone(), ws["keys"].view(B, Nk, C).clone()


def _check_shapes(C, T, Nk, NH):
    if C != 256 or NH != 8:
        raise NotImplementedError("the HIP two-way transformer is built for SAM's 256 channels and 8 heads (build_sam.py:84-90)")
    if T > 16:
        raise NotImplementedError("more than 11 sparse prompt tokens per prompt set")
    if Nk > 4096:
        raise NotImplementedError("more than 4096 image tokens")


def _shared_pe(pe, Nk):
    """[B or 1, Nk, 256] positional encoding of the image tokens -> the one [Nk,256] grid the kernels add per prompt set (SAM
    repeats one grid over the batch, mask_decoder.py:128)."""
    p = pe.reshape(-1, Nk, pe.shape[-1])
    return p[0].float().contiguous()


def _workspace(cache, B, T, Nk, dev, h16):
    key = (B, T, Nk, h16)
    if key not in cache:
        while len(cache) >= 4:          # every distinct number of prompt sets has its own ~13 MB-per-set workspace:
            cache.pop(next(iter(cache)))    # keep the four most recently created ones
        e = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        M = B * Nk
        pdt = torch.float16 if h16 else torch.float32
        cache[key] = dict(
            keys=e((M, 256)), k16=e((M, 256), torch.float16) if h16 else None,
            kpe16=e((M, 256), torch.float16) if h16 else None,
            p0=e((M, 128), pdt), p1=e((M, 128), pdt),
            q=e((B * T, 256)), tq128=e((B * T, 128)), ta128=e((B * T, 128)), tq=e((B * T, 256)), tk=e((B * T, 256)), tv=e((B * T, 256)),
            ta=e((B * T, 256)), t1=e((B * T, 256)), hid=e((B * T, 2048)), parts=e((8, B * T, 256)), t2i_part=e((16 * B * 8 * T * 18,)))
    return cache[key]


class TwoWayTransformer(nn.Module):
    def __init__(self, depth, embedding_dim, num_heads, mlp_dim, activation=nn.ReLU, attention_downsample_rate=2):
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate, (i == 0))
            for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)
        self._ws = {}

    def _apply(self, fn, *a, **k):
        self._ws = {}
        return super()._apply(fn, *a, **k)

    def run_tokens(self, feat_tok, pe_tok, tokens, dense_vec, img_of_prompt=None, image_side_fp16=False, image_side_x3=True):
        """The decoder's hot path (transformer.py:62-106 for B prompt sets): feat_tok fp32 [Nk,256] (one image) or [n_img,Nk,256]
        token-major image embeddings, pe_tok fp32 [Nk,256], tokens fp32 [B,T,256] (= point_embedding: queries AND query_pe),
        dense_vec fp32 [256] added to every image token (the no-mask embedding), img_of_prompt int32 [B] (which image each prompt
        set belongs to; None = image 0). -> (queries fp32 [B*T,256], keys fp32 [B*Nk,256]: views of the workspace, overwritten by
        the next call, and the workspace itself)."""
        B, T, C = tokens.shape
        Nk = feat_tok.shape[-2]
        _check_shapes(C, T, Nk, self.num_heads)
        ws = _workspace(self._ws, B, T, Nk, tokens.device, image_side_fp16)
        tok2 = tokens.reshape(B * T, 256).contiguous()  # query_pe (transformer.py:88-96)
        r = _Run(ws, B, T, Nk, self.num_heads, tok2, pe_tok, image_side_fp16, image_side_x3 and not image_side_fp16)
        ops.ln_pe(feat_tok, pe_tok, B * Nk, y32=ws["keys"], y16=ws["k16"], ype16=ws["kpe16"], add_vec=dense_vec, in_mod=Nk, pe_mod=Nk,
                  img_of_prompt=img_of_prompt)
        for li, layer in enumerate(self.layers):
            layer._run(r, tok2 if li == 0 else ws["q"])          # queries = point_embedding (transformer.py:85)
        if len(self.layers) == 0:
            ws["q"].copy_(tok2)
        fin = self.final_attn_token_to_image._pack(("kw", "vw"))
        r.t2i(fin, (f32(self.norm_final_attn.weight), f32(self.norm_final_attn.bias)))
        return ws["q"], ws["keys"], ws

    @torch.no_grad()
    def forward(self, image_embedding, image_pe, point_embedding):
        """transformer.py:62-106: image_embedding [B,C,h,w], image_pe [B or 1,C,h,w] (one grid shared by the batch), point_embedding
        [B,T,C] -> (queries [B,T,C], keys [B,h*w,C])."""
        B, C, h, w = image_embedding.shape
        Nk = h * w
        feat = image_embedding.flatten(2).permute(0, 2, 1).float().contiguous()
        pe = _shared_pe(image_pe.flatten(2).permute(0, 2, 1), Nk)
        dev = feat.device
        q, keys, _ = self.run_tokens(feat, pe, point_embedding.float().contiguous(), torch.zeros(C, dtype=torch.float32, device=dev),
                                     img_of_prompt=torch.arange(B, dtype=torch.int32, device=dev))
        T = point_embedding.shape[1]
        return q.view(B, T, C).clone(), keys.view(B, Nk, C).clone()
