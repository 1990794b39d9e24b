"""ALP module (`MultiProtoAsConv`) on the HIP kernels of csrc/alp.hip.

Mirrors /root/reference/models/alpmodule.py:21-198: same constructor, same `forward` arguments, modes
'mask' / 'gridconv' / 'gridconv+' and the same ValueError for anything else. The reference's NCHW tensors are
accepted at this boundary; internally features are token-major [h*w, C] (what the ViT emits), which is what
`FewShotSeg` feeds directly through `scores_token_major` without any permute.
Visualisation by-products of the reference (`debug_assign`, `vis_dict`, `proto_grid`; alpmodule.py:69-75,
121-128) are not produced: the tuple positions are kept and filled with None / {}.
"""
import torch
import torch.nn as nn

from . import ops

_MODES = {"mask": 0, "gridconv+": 1, "gridconv": 2}


def safe_norm_eps():
    return 1e-4  # alpmodule.py:14


class MultiProtoAsConv(nn.Module):
    def __init__(self, proto_grid, feature_hw, embed_dim=768, use_attention=False, upsample_mode="bilinear"):
        super().__init__()
        if use_attention:
            raise NotImplementedError("use_attention=True is never enabled by the reference (grid_proto_fewshot.py:117)")
        self.feature_hw = feature_hw
        self.proto_grid = proto_grid
        self.upsample_mode = upsample_mode
        self.kernel_size = [ft_l // grid_l for ft_l, grid_l in zip(feature_hw, proto_grid)]  # alpmodule.py:34
        self.embed_dim = embed_dim
        self._bank = None

    # ---- token-major core ------------------------------------------------------------------------------------
    def build_bank(self, sup_tok, ld, h, w, fg_mask, pool_w, thresh=0.95, force_mode=-1, bank=None, bg_mask=None):
        """sup_tok: fp32 token-major support features; fg_mask fp32 [MH,MW]. Builds bg ('gridconv' on 1-mask or
        bg_mask) and fg banks ('gridconv+' or 'mask' by the reference rule, grid_proto_fewshot.py:253-256)."""
        return ops.alp_bank(sup_tok, ld, h, w, self.embed_dim, fg_mask, pool_w, self.kernel_size[0], thresh,
                            safe_norm_eps(), bank=bank, force_mode=force_mode, bmask=bg_mask)

    def scores_token_major(self, qry_tok, q_bstride, ld, B, npix, bank, pred=None, which_only=-1):
        """-> fp32 [B, 2, npix]: (bg score, fg score) = sum_p softmax(20 cos) * 20 cos (alpmodule.py:67-70,79-82)."""
        return ops.alp_sim(qry_tok, q_bstride, ld, B, npix, self.embed_dim, bank, pred=pred, eps=safe_norm_eps(),
                           sim_scale=20.0, which_only=which_only)

    @staticmethod
    def merge_banks(banks, which):
        """One bank holding the background (which = 0) or foreground (1) prototypes of several banks, concatenated in order: what
        the reference's `get_prototypes` yields when `sup_x` carries several shots (alpmodule.py:111-131,155-158: the grid
        prototypes of all shots, for 'gridconv+' plus every shot's global prototype - a softmax-weighted sum does not care about
        the order). Host-side (one 8-int read per bank): banks are built once per support set, not per query."""
        key = ops.META_NFG if which else ops.META_NBG
        cnt = [int(m[key]) for m in torch.stack([b.meta for b in banks]).cpu()]
        tot = sum(cnt)
        merged = ops.AlpBank(max(tot, 1), banks[0].C, banks[0].bank.device)
        if tot:
            merged.bank[which * merged.cap:which * merged.cap + tot] = torch.cat(
                [b.bank[which * b.cap:which * b.cap + n] for b, n in zip(banks, cnt)], dim=0)
        merged.meta[key] = tot
        if which:
            merged.meta[ops.META_FGMODE] = banks[0].meta[ops.META_FGMODE]
        return merged

    # ---- reference-shaped API --------------------------------------------------------------------------------
    def forward(self, qry, sup_x, sup_y, mode, thresh, isval=False, val_wsize=None, vis_sim=False,
                get_prototypes=False, **kwargs):
        if mode not in _MODES:
            raise ValueError(f"Invalid mode: {mode}. Expected 'mask', 'gridconv', or 'gridconv+'.")
        qry = qry.squeeze(1)                      # [1, C, h, w]
        sup_x = sup_x.squeeze(0).squeeze(1)       # [nshot, C, h, w]
        sup_y = sup_y.squeeze(0)
        nshot = sup_x.shape[0]
        C, h, w = qry.shape[-3:]
        if val_wsize is None or not isval:
            pool_w = self.kernel_size[0]          # alpmodule.py:186-189 / avg_pool_op
        else:
            pool_w = val_wsize
        sup_y = sup_y.reshape(nshot, h, w).float().contiguous()
        qry_tok = qry[0].float().permute(1, 2, 0).reshape(h * w, C).contiguous()
        banks = []
        for i in range(nshot):
            sup_tok = sup_x[i].float().permute(1, 2, 0).reshape(h * w, C).contiguous()
            banks.append(self.build_bank(sup_tok, C, h, w, sup_y[i], pool_w, thresh, force_mode=_MODES[mode]))
        if mode == "mask":
            # one un-normalised prototype per shot, the prediction is the MAX of the shots' cosine maps (alpmodule.py:59-62)
            pred = None
            for b in banks:
                p = self.scores_token_major(qry_tok, h * w * C, C, 1, h * w, b, which_only=1)
                pred = p if pred is None else torch.maximum(pred, p)
        else:
            bank = banks[0] if nshot == 1 else self.merge_banks(banks, 1)
            if mode == "gridconv" and int(bank.meta[ops.META_NFG].item()) == 0:
                print("failed to find prototypes")
                raise RuntimeError("MultiProtoAsConv: no prototype passed the threshold (the reference fails in F.conv2d)")
            pred = self.scores_token_major(qry_tok, h * w * C, C, 1, h * w, bank, which_only=1)
        pred_grid = pred[:, 1].reshape(1, 1, h, w)
        return pred_grid, [None], {}, None
