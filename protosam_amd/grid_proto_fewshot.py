"""ALPNet (`FewShotSeg`) on the HIP path: DINOv2 encoder + ALP prototype matching.

Mirrors /root/reference/models/grid_proto_fewshot.py (FewShotSeg.__init__ :33-44, get_encoder :46-81,
get_features :83-103, get_cls :105-121, forward :150-290) for inference. Differences that do not change
results:
  * the encoder is our `DinoVisionTransformer` (hub key names) instead of `torch.hub.load`;
  * features stay token-major; `supp_fts` / `qry_fts` in the returned 7-tuple are zero-copy permuted views with
    the reference's shapes;
  * the support image's features and prototype bank are cached across calls while the support tensors are
    unchanged (the reference re-encodes the support on every slice, :181-184; same values).
    `cache_support=False` reproduces the reference's per-call cost.
`which_model = 'dlfcn_res101'` selects the ResNet-101 encoder of `protosam_amd/backbone.py` (feature map ceil(S/8)).
Checkpoints trained with `lora > 0` load through `collapse_lora_state_dict` (the low-rank update is folded into the base
weights; util/lora.py:638-672). Out of scope here (training only): alignLoss, dino losses, LoRA training.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .alpmodule import MultiProtoAsConv
from .dinov2 import DinoVisionTransformer

DEFAULT_FEATURE_SIZE = 32  # util/consts.py
FG_PROT_MODE = "gridconv+"
BG_PROT_MODE = "gridconv"
FG_THRESH = 0.95
BG_THRESH = 0.95

_HUB_NAME = {"dinov2_b14": "dinov2_vitb14", "dinov2_l14": "dinov2_vitl14", "dinov2_l14_reg": "dinov2_vitl14_reg"}


def collapse_lora_state_dict(sd, scale=1.0):
    """LoraInjectedLinear keys -> plain nn.Linear keys with the low-rank update folded in (fp32)."""
    if not any(k.endswith(".lora_up.weight") for k in sd):
        return sd
    out = {}
    for k, v in sd.items():
        if k.endswith(".linear.weight"):
            p = k[:-len("linear.weight")]
            up, down = sd[p + "lora_up.weight"].float(), sd[p + "lora_down.weight"].float()
            out[p + "weight"] = v.float() + scale * (up @ down)
        elif k.endswith(".linear.bias"):
            out[k[:-len("linear.bias")] + "bias"] = v
        elif k.endswith(".lora_up.weight") or k.endswith(".lora_down.weight") or ".selector." in k:
            continue
        else:
            out[k] = v
    return out


class FewShotSeg(nn.Module):
    def __init__(self, image_size, pretrained_path=None, cfg=None, cache_support=True):
        super().__init__()
        self.image_size = image_size
        self.pretrained_path = pretrained_path
        self.config = cfg or {"align": False, "debug": False}
        self.cache_support = cache_support
        self._is_vit = self.config["which_model"] in _HUB_NAME
        self.get_encoder()
        self.get_cls()
        self._sup_cache = []
        if self.pretrained_path:
            self.load_state_dict(torch.load(self.pretrained_path), strict=True)
            print(f"###### Pre-trained model f{self.pretrained_path} has been loaded ######")

    def get_encoder(self):
        which = self.config["which_model"]
        if which in _HUB_NAME:
            self.encoder = DinoVisionTransformer(_HUB_NAME[which], depth=self.config.get("encoder_depth"))
            s = max(self.image_size // 14, DEFAULT_FEATURE_SIZE)
            self.config["feature_hw"] = [s, s]
        elif which in ("dlfcn_res101", "default"):                              # :49-53
            from .backbone import TVDeeplabRes101Encoder
            self.encoder = TVDeeplabRes101Encoder(self.config.get("use_coco_init", False),
                                                  layers=self.config.get("resnet_layers", (3, 4, 23, 3)))
            s = math.ceil(self.image_size / 8)
            self.config["feature_hw"] = [s, s]
        else:
            raise NotImplementedError(f"Backbone network {which} not implemented")
        # lora > 0 (util/lora.py:258-312 replaces every nn.Linear of the DINOv2 blocks by LoraInjectedLinear): inference needs
        # no second path - `load_state_dict` folds `lora_up @ lora_down` into the base weight (collapse_lora, :638-672)

    def get_cls(self):
        proto_hw = self.config["proto_grid_size"]
        if self.config["cls_name"] != "grid_proto":
            raise NotImplementedError(f'Classifier {self.config["cls_name"]} not implemented')
        self.cls_unit = MultiProtoAsConv(proto_grid=[proto_hw, proto_hw], feature_hw=self.config["feature_hw"],
                                         embed_dim=self.encoder.embed_dim)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts the reference's checkpoints, including those trained with `lora > 0`: the injected layers' keys
        (`<p>.linear.weight`, `<p>.linear.bias`, `<p>.lora_down.weight`, `<p>.lora_up.weight`) are collapsed to
        `<p>.weight = W + scale * up @ down` (scale = 1.0, inject_trainable_lora's default; util/lora.py:34-59,638-672)."""
        return super().load_state_dict(collapse_lora_state_dict(state_dict), strict=strict, **kw)

    def _load_from_state_dict(self, *a, **k):
        # the cached support banks were built with the previous encoder weights (keyed on the input tensors only)
        self._sup_cache = []
        return super()._load_from_state_dict(*a, **k)

    # ---- features ------------------------------------------------------------------------------------------------
    def _grid(self):
        """(S, g): encoder input side and the side of the feature map the classifier sees (>= 32, :96-98)."""
        if not self._is_vit:
            return self.image_size, math.ceil(self.image_size / 8)
        S = self.image_size // 14 * 14
        return S, max(S // 14, DEFAULT_FEATURE_SIZE)

    def _patch_tokens(self, imgs):
        """imgs [B,3,H,W] -> token-major patch features fp32 [B, g*g, C] (view of a workspace, or the 32x32 bilinear
        upsample of it when the encoder yields fewer patches) plus (batch stride, row stride) in elements."""
        S, g = self._grid()
        if not self._is_vit:
            if self.config["which_model"] != "dlfcn_res101":                    # get_features has no 'default' branch (:99-101)
                raise NotImplementedError(f'Backbone network {self.config["which_model"]} not implemented')
            tok, h, w = self.encoder.forward_tokens(imgs.float())
            assert h == g and w == g, (h, w, g)
            return tok, g * g * self.encoder.embed_dim, self.encoder.embed_dim
        ge = S // 14
        C = self.encoder.embed_dim
        R = self.encoder.num_register_tokens
        t = self.encoder.forward_tokens(imgs.float(), S)           # [B, 1+R+ge*ge, C]
        B, N, _ = t.shape
        tok = t[:, 1 + R:]
        if ge == g:
            return tok, N * C, C
        up = ops.bilinear_tokens(tok, N * C, C, B, ge, ge, C, g, g)
        return up, g * g * C, C

    def get_features(self, imgs_concat):
        """-> [B, C, h, w] as the reference (a permuted view of the token-major buffer)."""
        S, g = self._grid()
        tok, _, C = self._patch_tokens(imgs_concat)
        B = tok.shape[0]
        return tok.reshape(B, g, g, C).permute(0, 3, 1, 2)

    def _support_bank(self, supp, fg, bg, pool_w):
        """Support features + prototype bank, cached while the support is unchanged (a small LRU: one entry per z-part
        of a scan, validation_protosam.py:355-362). Entries hold references to the tensors they were built from (so
        their storage cannot be recycled under the same address); a different tensor object with identical content
        (the reference caller re-uploads the support every slice, validation_protosam.py:374-385) is recognised by an
        on-device equality test instead of a re-encode."""
        if self.cache_support:
            epoch = getattr(self.encoder, "_weights_epoch", 0)   # bumped whenever the encoder's weights are (re)loaded / moved
            for i, (k_supp, k_fg, k_bg, k_pool, k_ver, bank, tok) in enumerate(self._sup_cache):
                if k_pool != (pool_w, epoch) or k_supp.shape != supp.shape or k_fg.shape != fg.shape:
                    continue
                same_obj = (k_supp is supp and k_fg is fg and k_ver == (supp._version, fg._version))
                if same_obj or (k_supp is not supp and torch.equal(k_supp, supp) and torch.equal(k_fg, fg)
                                and (k_bg is bg or (k_bg is not None and bg is not None and torch.equal(k_bg, bg)))):
                    if i:
                        self._sup_cache.insert(0, self._sup_cache.pop(i))
                    return bank, tok
        S, g = self._grid()
        C = self.encoder.embed_dim
        t, _, _ = self._patch_tokens(supp)
        tok = t[0].clone()                                        # [g*g, C] kept for the cache / returned tuple
        fg2 = fg.reshape(fg.shape[-2], fg.shape[-1]).float().contiguous()
        bg2 = None
        if bg is not None and not bool(torch.equal(bg.float(), 1 - fg.float())):   # ProtoSAM.py:63 builds 1 - fg
            bg2 = bg.reshape(bg.shape[-2], bg.shape[-1]).float().contiguous()
        bank = self.cls_unit.build_bank(tok, C, g, g, fg2, pool_w, FG_THRESH, force_mode=-1, bank=None, bg_mask=bg2)
        if self.cache_support:
            self._sup_cache.insert(0, (supp, fg, bg, (pool_w, getattr(self.encoder, "_weights_epoch", 0)),
                                       (supp._version, fg._version), bank, tok))
            del self._sup_cache[16:]          # (3 z-parts x a few shots)
        return bank, tok

    def _merged_bg_bank(self, banks):
        """n_shots > 1: the background classifier sees every shot's grid prototypes at once, concatenated in shot order
        (grid_proto_fewshot.py:239-240 hands all shots to one `cls_unit` call; alpmodule.py:111-131). Built once per set of shot
        banks (one 8-int read per shot to size it) and cached beside them."""
        key = tuple(id(b) for b in banks)
        hit = self.__dict__.get("_ms_cache")
        if hit is not None and hit[0] == key:
            return hit[2]
        merged = self.cls_unit.merge_banks(banks, 0)
        self._ms_cache = (key, list(banks), merged)      # (holds the shot banks: their ids stay unique while this entry lives)
        return merged

    def _shot_banks(self, supp_imgs, fore_mask, back_mask, pool_w):
        """(bank, support tokens) per shot of a 1-way support set. Encodes whatever is not cached - through the encoder's token
        workspace, so call it BEFORE the query is encoded."""
        if len(supp_imgs) != 1:
            raise AssertionError("Multi-shot has not been implemented yet")      # (the reference's message for n_ways != 1, :171)
        shots = supp_imgs[0]
        if any(s.shape[0] != 1 for s in shots):
            raise NotImplementedError("support batch size 1 per shot (validation_protosam.py:346-362)")
        return [self._support_bank(shots[i], fore_mask[0][i], back_mask[0][i], pool_w) for i in range(len(shots))]

    def _match(self, qry_tok, q_bstride, q_ld, n, pairs):
        """prototype match of n query slices against one support set (`_shot_banks`) -> scores [n, 2, g*g]. One shot: one bank,
        one launch. Several (grid_proto_fewshot.py:244-266): background against the merged bank, foreground per shot with that
        shot's own mode, element-wise max over the shots."""
        S, g = self._grid()
        if len(pairs) == 1:
            bank = pairs[0][0]
            pred = self.cls_unit.scores_token_major(qry_tok, q_bstride, q_ld, n, g * g, bank)   # [n, 2, g*g]
            self._check_bank(bank)
            return pred
        banks = [p[0] for p in pairs]
        merged = self._merged_bg_bank(banks)
        pred = self.cls_unit.scores_token_major(qry_tok, q_bstride, q_ld, n, g * g, merged, which_only=0)
        self._check_bank(merged)
        tmp = None
        for i, bank in enumerate(banks):
            tmp = self.cls_unit.scores_token_major(qry_tok, q_bstride, q_ld, n, g * g, bank, pred=tmp, which_only=1)
            if i == 0:
                pred[:, 1].copy_(tmp[:, 1])
            else:
                torch.maximum(pred[:, 1], tmp[:, 1], out=pred[:, 1])
        return pred

    def forward(self, supp_imgs, fore_mask, back_mask, qry_imgs, isval, val_wsize, show_viz=False, supp_fts=None):
        n_ways, n_queries = len(supp_imgs), len(qry_imgs)
        assert n_ways == 1, "Multi-shot has not been implemented yet"
        assert n_queries == 1
        if supp_fts is not None:
            raise NotImplementedError("supp_fts is unusable in the reference as well (SURVEY Q18)")
        qry = qry_imgs[0]
        img_size = supp_imgs[0][0].shape[-2:]
        S, g = self._grid()
        C = self.encoder.embed_dim
        pool_w = val_wsize if (isval and val_wsize is not None) else self.cls_unit.kernel_size[0]
        pairs = self._shot_banks(supp_imgs, fore_mask, back_mask, pool_w)
        sup_tok = pairs[0][1]
        qry_tok, q_bstride, q_ld = self._patch_tokens(qry)        # [B, g*g, C] (strided view or upsampled copy)
        B = qry_tok.shape[0]
        pred = self._match(qry_tok, q_bstride, q_ld, B, pairs)
        output = ops.bilinear_nchw(pred.view(B, 2, g, g), img_size[0], img_size[1])   # :272-273
        supp_view = sup_tok.reshape(1, 1, 1, g, g, C).permute(0, 1, 2, 5, 3, 4)
        qry_view = qry_tok.unflatten(1, (g, g)).unsqueeze(0).permute(0, 1, 4, 2, 3)   # zero-copy [1,B,C,g,g]
        return output, 0.0, [None, None], None, None, supp_view, qry_view

    @torch.no_grad()
    def forward_groups(self, qry_imgs, groups):
        """MI355X extension: ONE encoder forward for a batch of query slices that belong to DIFFERENT support sets (a rank of a
        multi-GPU job, or a batch that straddles two z-parts of a scan: validation_protosam.py:352-388 changes the support with the
        part). qry_imgs [B,3,H,W]; groups: list of (supp_imgs, fore_mask, back_mask, isval, val_wsize, n) in batch order, the
        first n slices matched against the first support set and so on (sum n == B). Only the prototype match depends on the
        support: every group's logits equal `forward(...)` of that group alone. Returns logits [B,2,H,W]."""
        S, g = self._grid()
        B = qry_imgs.shape[0]
        assert sum(gr[-1] for gr in groups) == B
        img_size = groups[0][0][0][0].shape[-2:]
        allpairs = []                      # every group's banks first: a support that is not cached yet is encoded through the same
        for supp_imgs, fore_mask, back_mask, isval, val_wsize, n in groups:       # token workspace the query's tokens will live in
            assert len(supp_imgs) == 1, "Multi-shot has not been implemented yet"
            assert tuple(supp_imgs[0][0].shape[-2:]) == tuple(img_size)
            pool_w = val_wsize if (isval and val_wsize is not None) else self.cls_unit.kernel_size[0]
            allpairs.append(self._shot_banks(supp_imgs, fore_mask, back_mask, pool_w))
        qry_tok, q_bstride, q_ld = self._patch_tokens(qry_imgs)
        out = torch.empty((B, 2, img_size[0], img_size[1]), dtype=torch.float32, device=qry_imgs.device)
        i0 = 0
        for pairs, gr in zip(allpairs, groups):
            n = gr[-1]
            pred = self._match(qry_tok[i0:i0 + n], q_bstride, q_ld, n, pairs)   # [n, 2, g*g]
            ops.bilinear_nchw(pred.view(n, 2, g, g), img_size[0], img_size[1], out=out[i0:i0 + n])
            i0 += n
        return out

    @torch.no_grad()
    def forward_classes(self, supp_img, fore_masks, qry_img, isval=True, val_wsize=None):
        """Several 1-way episodes on the SAME support / query image pair (the multi-class loop of /root/reference/validation.py:207,
        BASELINE config 5: one prototype bank per class): the support and the query image are each encoded ONCE and the query
        tokens are matched against every class's bank. fore_masks: list of [1,H,W] masks. Returns a list of logits [1,2,H,W],
        each identical to `forward(...)` of that class alone."""
        S, g = self._grid()
        C = self.encoder.embed_dim
        img_size = supp_img.shape[-2:]
        pool_w = val_wsize if (isval and val_wsize is not None) else self.cls_unit.kernel_size[0]
        key = (supp_img, supp_img._version, tuple((m, m._version) for m in fore_masks), pool_w, getattr(self.encoder, "_weights_epoch", 0))
        hit = getattr(self, "_cls_cache", None)
        if (self.cache_support and hit is not None and hit[0][0] is supp_img and hit[0][1] == key[1] and hit[0][3:] == key[3:]
                and len(hit[0][2]) == len(key[2]) and all(a[0] is b[0] and a[1] == b[1] for a, b in zip(hit[0][2], key[2]))):
            banks = hit[1]
        else:
            t, _, _ = self._patch_tokens(supp_img)
            tok = t[0].clone()
            banks = []
            for fg in fore_masks:
                fg2 = fg.reshape(fg.shape[-2], fg.shape[-1]).float().contiguous()
                banks.append(self.cls_unit.build_bank(tok, C, g, g, fg2, pool_w, FG_THRESH, force_mode=-1, bank=None, bg_mask=None))
            self._cls_cache = (key, banks)
        qry_tok, q_bstride, q_ld = self._patch_tokens(qry_img)
        B = qry_tok.shape[0]
        outs = []
        for bank in banks:
            pred = self.cls_unit.scores_token_major(qry_tok, q_bstride, q_ld, B, g * g, bank)
            self._check_bank(bank)
            outs.append(ops.bilinear_nchw(pred.view(B, 2, g, g), img_size[0], img_size[1]))
        return outs

    def _check_bank(self, bank):
        """The reference raises inside F.conv2d when a bank is empty (alpmodule.py:193-196). One 8-int D2H read."""
        if self.config.get("skip_bank_check", False):
            return
        if bank.__dict__.get("_checked"):
            return
        m = bank.meta.cpu()
        if int(m[ops.META_NBG]) == 0:
            print("failed to find prototypes")
            raise RuntimeError("no background prototype passed the 0.95 coverage threshold")
        bank._checked = True
