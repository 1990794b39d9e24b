"""SAM mask decoder on the HIP kernels (names of models/segment_anything/modeling/mask_decoder.py:16-176).

`predict_masks_tokens` restates mask_decoder.py:112-149 for a batch of B prompt sets on one or several images: the two-way
transformer (`TwoWayTransformer.run_tokens`, transformer.py) and then ConvTranspose(2,2) #1 as a GEMM at fp32 accuracy
(`psam_gemm_f32x3` / `psam_gemm_f32`) - with fp16 operands the image side alone measured 0.8e-3 .. 1.2e-3 on
`sigmoid(low_res_masks)`; that path is kept behind `image_side_fp16 = True` for throughput experiments with hundreds of prompt
sets. The rest of the upscaling and the hyper-network product are fused in `upscale_tail` so `upscaled_embedding` [B,32,256,256]
is never materialised.
"""
import os

import torch
import torch.nn as nn

from ... import ops
from .common import LayerNorm2d, f16, f32
from .transformer import TwoWayTransformer  # noqa: F401


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
    def _packed(self):
        """The decoder's own weights (the two-way transformer packs its own: transformer.py)."""
        if self._cache is not None:
            return self._cache
        pk = {}
        pk["out_tok"] = torch.cat([f32(self.iou_token.weight), f32(self.mask_tokens.weight)], 0).contiguous()
        up0, ln, up3 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
        # ConvTranspose2d weight [in, out, kh, kw]: GEMM row (dy*2+dx)*64 + co  <-  W[:, co, dy, dx]
        pk["up1_w16"] = f16(up0.weight.permute(2, 3, 1, 0).reshape(4 * 64, 256))
        pk["up1_w"] = f32(up0.weight.permute(2, 3, 1, 0).reshape(4 * 64, 256))
        sc = ops.split_scale_for(pk["up1_w"], ops.X3_WEIGHT_SCALE)
        pk["up1_wx3"] = ops.split_weight_f16(pk["up1_w"], sc) + (sc,)
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
        key = (B, T, Nk)
        if key not in self._ws:
            while len(self._ws) >= 4:          # (the transformer keeps the large per-shape buffers; these are ~1 MB per prompt set)
                self._ws.pop(next(iter(self._ws)))
            e = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
            self._ws[key] = dict(u1=e((B * Nk, 256)), h1=e((B, 4, 256)), h2=e((B, 4, 256)), hyper=e((B, 4, 32)), i1=e((B, 256)),
                                 i2=e((B, 256)), iou=e((B, 4)), masks=e((B, 4, 256, 256)))
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
        lin = ops.small_linear
        h16 = self.image_side_fp16
        x3 = self.image_side_x3 and not h16
        # the two-way transformer (mask_decoder.py:131, transformer.py:62-106)
        q, keys, tws = self.transformer.run_tokens(feat_tok, pe_tok, tokens, dense_vec, img_of_prompt=img_of_prompt,
                                                   image_side_fp16=h16, image_side_x3=x3)
        hs = q.view(B, T, 256)
        # output upscaling + hyper-networks (mask_decoder.py:137-144) and IoU head (:147)
        if h16:
            ops.gemm(tws["k16"], pk["up1_w16"], pk["up1_b"], out=ws["u1"], epilogue=ops.EPI_F32)
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
