"""SAM mask decoder on the HIP kernels (names of models/segment_anything/modeling/mask_decoder.py:16-176).

`predict_masks_tokens` restates mask_decoder.py:112-149 + transformer.py:62-106,151-182 for a batch of B prompt sets on
ONE image. Token side (T <= 16 tokens/prompt): fp32 `small_linear` / `small_attention`. Image side (4096 tokens):
k/v/q projections, the i2t out-projection and ConvTranspose(2,2) #1 are GEMMs on the exact-fp32 MFMA (`psam_gemm_f32`,
`keys + pe` added on the operand load) - 3.2 GFLOP per prompt set, ~25 us - so that `sigmoid(low_res_masks)` keeps a
margin under the 1e-3 parity bound (with fp16 operands this stage alone measured 0.8e-3 .. 1.2e-3; that path is kept behind
`image_side_fp16 = True` for throughput experiments with hundreds of prompt sets). Token->image attention is
`t2i_attention`; the rest of the upscaling and the hyper-network product are fused in `upscale_tail` so
`upscaled_embedding` [B,32,256,256] is never materialised.
"""
import os

import torch
import torch.nn as nn

from ... import ops
from .common import LayerNorm2d, f16, f32
from .transformer import TwoWayTransformer  # noqa: F401

LN_EPS = 1e-5  # nn.LayerNorm default (transformer.py:133-144)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, sigmoid_output=False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output
        if sigmoid_output:
            raise NotImplementedError("sigmoid_output is never enabled by SAM (mask_decoder.py:61-69)")


class MaskDecoder(nn.Module):
    def __init__(self, *, transformer_dim, transformer, num_multimask_outputs=3, activation=nn.GELU, iou_head_depth=3,
                 iou_head_hidden_dim=256):
        super().__init__()
        assert transformer_dim == 256 and num_multimask_outputs == 3 and iou_head_depth == 3
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4), activation(),
            nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2), activation())
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)
        self._cache = None
        self._ws = {}
        self.image_side_fp16 = False   # True: fp16-operand MFMA GEMMs on the image side (faster for >= 256 prompt sets, ~1e-3)
        # the image side's fp32 products as three fp16 MFMA products on (hi, lo) halves (ops.gemm_f32x3: fp32 accuracy to 2^-22 per term,
        # the fp16 matrix pipe instead of the 16x slower fp32 one); PSAM_DECODER_X3=0 / `image_side_x3 = False`: the exact-fp32 MFMA kernel
        self.image_side_x3 = os.environ.get("PSAM_DECODER_X3", "1") != "0"

    def _apply(self, fn, *a, **k):
        self._cache, self._ws = None, {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load
        self._cache = None
        return super()._load_from_state_dict(*a, **k)

    # ---- packing -----------------------------------------------------------------------------------------------
    @staticmethod
    def _attn_pack(a, image_side):
        d = dict(qw=f32(a.q_proj.weight), qb=f32(a.q_proj.bias), kw=f32(a.k_proj.weight), kb=f32(a.k_proj.bias),
                 vw=f32(a.v_proj.weight), vb=f32(a.v_proj.bias), ow=f32(a.out_proj.weight), ob=f32(a.out_proj.bias))
        mods = {"qw": a.q_proj, "kw": a.k_proj, "vw": a.v_proj, "ow": a.out_proj}
        for nm in image_side:  # projections applied to the 4096 image tokens run as fp16 MFMA GEMMs
            d[nm + "16"] = f16(mods[nm].weight)
            d[nm + "x3"] = ops.split_weight_f16(mods[nm].weight, ops.X3_WEIGHT_SCALE) + (ops.X3_WEIGHT_SCALE,)      # (hi, lo, scale) for ops.gemm_f32x3
        return d

    def _packed(self):
        if self._cache is not None:
            return self._cache
        tr = self.transformer
        pk = dict(layers=[])
        for L in tr.layers:
            pk["layers"].append(dict(
                sa=self._attn_pack(L.self_attn, ()), t2i=self._attn_pack(L.cross_attn_token_to_image, ("kw", "vw")),
                i2t=self._attn_pack(L.cross_attn_image_to_token, ("qw", "ow")),
                n=[(f32(n.weight), f32(n.bias)) for n in (L.norm1, L.norm2, L.norm3, L.norm4)],
                l1w=f32(L.mlp.lin1.weight), l1b=f32(L.mlp.lin1.bias), l2w=f32(L.mlp.lin2.weight),
                l2b=f32(L.mlp.lin2.bias), skip_pe=L.skip_first_layer_pe))
        pk["final"] = self._attn_pack(tr.final_attn_token_to_image, ("kw", "vw"))
        pk["nf"] = (f32(tr.norm_final_attn.weight), f32(tr.norm_final_attn.bias))
        pk["out_tok"] = torch.cat([f32(self.iou_token.weight), f32(self.mask_tokens.weight)], 0).contiguous()
        up0, ln, up3 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
        # ConvTranspose2d weight [in, out, kh, kw]: GEMM row (dy*2+dx)*64 + co  <-  W[:, co, dy, dx]
        pk["up1_w16"] = f16(up0.weight.permute(2, 3, 1, 0).reshape(4 * 64, 256))
        pk["up1_w"] = f32(up0.weight.permute(2, 3, 1, 0).reshape(4 * 64, 256))
        pk["up1_wx3"] = ops.split_weight_f16(pk["up1_w"], ops.X3_WEIGHT_SCALE) + (ops.X3_WEIGHT_SCALE,)
        pk["up1_b"] = f32(up0.bias).repeat(4).contiguous()
        pk["up_lnw"], pk["up_lnb"] = f32(ln.weight), f32(ln.bias)
        pk["up2_w"] = f32(up3.weight.permute(0, 2, 3, 1).reshape(64, 4 * 32))  # [c, (dy2*2+dx2)*32 + c2]
        pk["up2_b"] = f32(up3.bias)
        hm = self.output_hypernetworks_mlps
        pk["hyp_w"] = [torch.stack([f32(m.layers[i].weight) for m in hm]).contiguous() for i in range(3)]
        pk["hyp_b"] = [torch.stack([f32(m.layers[i].bias) for m in hm]).contiguous() for i in range(3)]
        pk["iou"] = [(f32(l.weight), f32(l.bias)) for l in self.iou_prediction_head.layers]
        self._cache = pk
        return pk

    def _workspace(self, B, T, Nk, dev):
        h = self.image_side_fp16
        key = (B, T, h)
        if key not in self._ws:
            while len(self._ws) >= 4:          # every distinct number of prompt sets has its own ~13 MB-per-set workspace:
                self._ws.pop(next(iter(self._ws)))   # keep the four most recently created ones
            e = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
            M = B * Nk
            pdt = torch.float16 if h else torch.float32
            self._ws[key] = dict(
                keys=e((M, 256)), k16=e((M, 256), torch.float16) if h else None,
                kpe16=e((M, 256), torch.float16) if h else None,
                p0=e((M, 128), pdt), p1=e((M, 128), pdt), u1=e((M, 256)),
                q=e((B * T, 256)), tq128=e((B * T, 128)), ta128=e((B * T, 128)), tq=e((B * T, 256)), tk=e((B * T, 256)), tv=e((B * T, 256)), ta=e((B * T, 256)),
                t1=e((B * T, 256)), hid=e((B * T, 2048)), parts=e((8, B * T, 256)), h1=e((B, 4, 256)), h2=e((B, 4, 256)), hyper=e((B, 4, 32)),
                i1=e((B, 256)), i2=e((B, 256)), iou=e((B, 4)), masks=e((B, 4, 256, 256)), t2i_part=e((16 * B * 8 * T * 18,)))
        return self._ws[key]

    # ---- the decoder ---------------------------------------------------------------------------------------------
    def predict_masks_tokens(self, feat_tok, pe_tok, tokens, dense_vec, img_of_prompt=None, masks_out=None,
                             iou_out=None):
        """feat_tok fp32 [Nk,256] (one image) or [n_img,Nk,256] token-major image embeddings, pe_tok fp32 [Nk,256],
        tokens fp32 [B,T,256] (output tokens ++ sparse prompts), dense_vec fp32 [256] (no-mask embedding),
        img_of_prompt int32 [B] (which image each prompt set belongs to; None = image 0).
        masks_out / iou_out: optional contiguous destinations (else views of the workspace, overwritten by the next call).
        -> masks [B,4,256,256], iou [B,4]."""
        pk = self._packed()
        B, T, _ = tokens.shape
        Nk = feat_tok.shape[-2]
        g = int(Nk ** 0.5)
        if T > 16:
            raise NotImplementedError("more than 11 sparse prompt tokens per prompt set")
        ws = self._workspace(B, T, Nk, tokens.device)
        NH = self.transformer.num_heads
        keys, k16, kpe16, q = ws["keys"], ws["k16"], ws["kpe16"], ws["q"]
        tok2 = tokens.reshape(B * T, 256).contiguous()  # query_pe (transformer.py:88-96)
        lin = ops.small_linear
        h16 = self.image_side_fp16
        x3 = self.image_side_x3 and not h16
        ops.ln_pe(feat_tok, pe_tok, B * Nk, y32=keys, y16=k16, ype16=kpe16, add_vec=dense_vec, in_mod=Nk, pe_mod=Nk,
                  img_of_prompt=img_of_prompt)

        def img_proj(w_name, ap, out, with_pe, heads=None):
            """image-token projection: (keys [+ key_pe]) @ W^T + b (transformer.py:228-230 of the 4096-token operand); heads = (Nk, hd):
            written head-major [B][NH][Nk][hd] for the token-to-image attention kernel"""
            if h16:
                ops.gemm(kpe16 if with_pe else k16, ap[w_name + "16"], ap[w_name[0] + "b"], out=out, epilogue=ops.EPI_F16)
            elif x3:
                ops.gemm_f32x3(keys, ap[w_name + "x3"], ap[w_name[0] + "b"], out=out, a2=pe_tok if with_pe else None, a2_mod=Nk, heads=heads)
            else:
                ops.gemm_f32(keys, ap[w_name], ap[w_name[0] + "b"], out=out, a2=pe_tok if with_pe else None, a2_mod=Nk, heads=heads)

        hm = (not h16) and T <= 16 and Nk >= 64       # K / V of the token-to-image attention head-major (contiguous 64-byte key rows)
        t2i_s = ops.t2i_split(B, NH, T, Nk) if NH == 8 else 1     # key ranges per (prompt set, head): few prompt sets leave the CUs idle

        def t2i(ap, resid_ln):
            lin(q, ap["qw"], ap["qb"], out=ws["tq128"], x2=tok2)
            img_proj("kw", ap, ws["p0"], True, heads=(Nk, 128 // NH) if hm else None)
            img_proj("vw", ap, ws["p1"], False, heads=(Nk, 128 // NH) if hm else None)
            ops.t2i_attention(ws["tq128"], ws["p0"], ws["p1"], ws["ta128"], B, T, Nk, NH, head_major=hm,
                              split=(t2i_s, ws["t2i_part"]))
            lin(ws["ta128"], ap["ow"], ap["ob"], out=ws["t1"], resid=q)
            ops.layernorm(ws["t1"], resid_ln[0], resid_ln[1], LN_EPS, out=q, out_dtype=torch.float32)

        for li, L in enumerate(pk["layers"]):
            sa = L["sa"]
            x2 = None if L["skip_pe"] else tok2
            if li == 0:
                src = tok2          # queries = point_embedding (transformer.py:85)
            else:
                src = q
            lin(src, sa["qw"], sa["qb"], out=ws["tq"], x2=x2)
            lin(src, sa["kw"], sa["kb"], out=ws["tk"], x2=x2)
            lin(src, sa["vw"], sa["vb"], out=ws["tv"])
            ops.small_attention(ws["tq"], ws["tk"], ws["tv"], ws["ta"], B, T, T, NH, 256 // NH, 256, 256, 256, 256)
            lin(ws["ta"], sa["ow"], sa["ob"], out=ws["t1"], resid=None if L["skip_pe"] else src)
            ops.layernorm(ws["t1"], L["n"][0][0], L["n"][0][1], LN_EPS, out=q, out_dtype=torch.float32)
            t2i(L["t2i"], L["n"][1])
            lin(q, L["l1w"], L["l1b"], out=ws["hid"], act=1)
            if B * T >= 32:      # 2048 -> 256 on few rows: eight K ranges side by side (psam_small_linear_splitk)
                ops.small_linear_splitk(ws["hid"], L["l2w"], L["l2b"], q, ws["t1"], ws["parts"], 8)
            else:
                lin(ws["hid"], L["l2w"], L["l2b"], out=ws["t1"], resid=q)
            ops.layernorm(ws["t1"], L["n"][2][0], L["n"][2][1], LN_EPS, out=q, out_dtype=torch.float32)
            ia = L["i2t"]
            img_proj("qw", ia, ws["p0"], True)
            lin(q, ia["kw"], ia["kb"], out=ws["tk"][:, :128], x2=tok2)
            lin(q, ia["vw"], ia["vb"], out=ws["tv"][:, :128])
            ops.small_attention(ws["p0"], ws["tk"][:, :128], ws["tv"][:, :128], ws["p1"], B, Nk, T, NH, 128 // NH, 128,
                                256, 256, 128)
            if h16:
                ops.gemm(ws["p1"], ia["ow16"], ia["ob"], out=keys, epilogue=ops.EPI_F32, resid=keys)
            elif x3:
                ops.gemm_f32x3(ws["p1"], ia["owx3"], ia["ob"], out=keys, resid=keys)
            else:
                ops.gemm_f32(ws["p1"], ia["ow"], ia["ob"], out=keys, resid=keys)
            ops.ln_pe(keys, pe_tok, B * Nk, y32=keys, y16=k16, ype16=kpe16, w=L["n"][3][0], b=L["n"][3][1], pe_mod=Nk,
                      eps=LN_EPS)
        t2i(pk["final"], pk["nf"])
        hs = q.view(B, T, 256)
        # output upscaling + hyper-networks (mask_decoder.py:137-144) and IoU head (:147)
        if h16:
            ops.gemm(k16, pk["up1_w16"], pk["up1_b"], out=ws["u1"], epilogue=ops.EPI_F32)
        elif x3:
            ops.gemm_f32x3(keys, pk["up1_wx3"], pk["up1_b"], out=ws["u1"])
        else:
            ops.gemm_f32(keys, pk["up1_w"], pk["up1_b"], out=ws["u1"])
        hx = hs[:, 1:5]
        lin(hx, pk["hyp_w"][0], pk["hyp_b"][0], out=ws["h1"], act=1, G=4, M=B, N=256, K=256, xg=256, wg=256 * 256,
            bg=256, yg=256, ldx=T * 256, ldy=4 * 256)
        lin(ws["h1"], pk["hyp_w"][1], pk["hyp_b"][1], out=ws["h2"], act=1, G=4, M=B, N=256, K=256, xg=256,
            wg=256 * 256, bg=256, yg=256, ldx=4 * 256, ldy=4 * 256)
        lin(ws["h2"], pk["hyp_w"][2], pk["hyp_b"][2], out=ws["hyper"], G=4, M=B, N=32, K=256, xg=256, wg=32 * 256,
            bg=32, yg=32, ldx=4 * 256, ldy=4 * 32)
        masks = ws["masks"] if masks_out is None else masks_out
        iou = ws["iou"] if iou_out is None else iou_out
        assert masks.is_contiguous() and iou.is_contiguous() and masks.shape[0] == B and iou.shape[0] == B
        ops.upscale_tail(ws["u1"], pk["up_lnw"], pk["up_lnb"], pk["up2_w"], pk["up2_b"], ws["hyper"], B, g, masks=masks)
        lin(hs[:, 0], pk["iou"][0][0], pk["iou"][0][1], out=ws["i1"], act=1, M=B, N=256, K=256, ldx=T * 256, ldy=256)
        lin(ws["i1"], pk["iou"][1][0], pk["iou"][1][1], out=ws["i2"], act=1)
        lin(ws["i2"], pk["iou"][2][0], pk["iou"][2][1], out=iou)
        return masks, iou, hs

    def build_tokens(self, sparse):
        """cat(iou_token, mask_tokens) ++ sparse prompts (mask_decoder.py:121-123)."""
        pk = self._packed()
        B = sparse.shape[0]
        return torch.cat([pk["out_tok"].unsqueeze(0).expand(B, -1, -1), sparse.float()], dim=1).contiguous()

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output):
        if image_embeddings.shape[0] != 1:
            raise NotImplementedError("one image per call (the predictor's usage, predictor.py:229-235)")
        d = dense_prompt_embeddings
        feat = image_embeddings[0].permute(1, 2, 0).reshape(-1, 256).float().contiguous()
        pe = image_pe[0].permute(1, 2, 0).reshape(-1, 256).float().contiguous()
        tokens = self.build_tokens(sparse_prompt_embeddings)
        if all(d.stride(i) == 0 or d.shape[i] == 1 for i in (0, 2, 3)):   # the broadcast no-mask embedding
            masks, iou, _ = self.predict_masks_tokens(feat, pe, tokens, d[0, :, 0, 0].float().contiguous())
        else:                                                             # mask prompts: one dense map per prompt set
            B = tokens.shape[0]
            if d.shape[0] != B:
                raise ValueError("dense prompt embeddings must have one map per prompt set")
            src = feat.unsqueeze(0) + d.permute(0, 2, 3, 1).reshape(B, -1, 256).float()   # mask_decoder.py:126-127
            masks, iou, _ = self.predict_masks_tokens(src.contiguous(), pe, tokens,
                                                      torch.zeros(256, dtype=torch.float32, device=feat.device),
                                                      img_of_prompt=torch.arange(B, dtype=torch.int32, device=feat.device))
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl], iou[:, sl]
