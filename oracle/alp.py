"""Oracle: ALPNet coarse segmenter (fp32, CPU). Test infrastructure only (see oracle/__init__.py).

Follows models/alpmodule.py (safe_norm :14-18, MultiProtoAsConv.get_prototypes :97-159,
get_prediction_from_prototypes :57-94, forward :161-198) and models/grid_proto_fewshot.py
(FewShotSeg.get_features :83-103, forward :150-290; FG/BG modes and thresholds :16-24) for the
inference configuration the caller uses (isval=True, val_wsize=2, n_ways=n_shots=n_queries=1,
validation_protosam.py:374-388), and for n_shots > 1 (`fewshot_*_multishot`, pinned by oracle/make_multishot_golden.py).
"""
import torch
import torch.nn.functional as F

FG_THRESH = BG_THRESH = 0.95   # grid_proto_fewshot.py:23-24
DEFAULT_FEATURE_SIZE = 32      # util/consts.py
SIM_SCALE = 20.0               # alpmodule.py:59,68,80


def safe_norm(x, dim=1, eps=1e-4):
    """alpmodule.py:14-18: x / max(||x||_2, eps) along `dim`."""
    n = torch.norm(x, p=2, dim=dim, keepdim=True)
    return x / torch.clamp(n, min=eps)


def masked_average(sup_x, sup_y):
    """alpmodule.py:99-100,155-156: sum(x*y)/(sum(y)+1e-5) over the map -> [nshot, C]."""
    return (sup_x * sup_y).sum(dim=(-1, -2)) / (sup_y.sum(dim=(-1, -2)) + 1e-5)


def get_prototypes(sup_x, sup_y, mode, val_wsize, thresh):
    """sup_x [nshot,C,h,w], sup_y [nshot,1,h,w] -> prototype bank [P,C] (L2-normalised for grid modes)."""
    if mode == "mask":
        return masked_average(sup_x, sup_y)  # un-normalised (alpmodule.py:102-103)
    nshot, C = sup_x.shape[:2]
    pooled = F.avg_pool2d(sup_x, val_wsize).view(nshot, C, -1).permute(0, 2, 1).reshape(-1, C)  # :111-113 row-major cells
    cover = F.avg_pool2d(sup_y, val_wsize).view(nshot, -1).reshape(-1)                          # :115,130
    protos = pooled[cover > thresh]                                                            # :131 strict >
    if mode == "gridconv":
        return safe_norm(protos)
    if mode == "gridconv+":
        return safe_norm(torch.cat([protos, masked_average(sup_x, sup_y)], dim=0))            # :155-158 global last
    raise ValueError(f"Invalid mode: {mode}. Expected 'mask', 'gridconv', or 'gridconv+'.")


def predict(protos, qry, mode):
    """qry [1,C,h,w]; returns [1,1,h,w] (alpmodule.py:57-94)."""
    if mode == "mask":
        sim = F.cosine_similarity(qry, protos[..., None, None], dim=1, eps=1e-4) * SIM_SCALE
        return sim.max(dim=0)[0][None, None]
    qn = safe_norm(qry)                                           # alpmodule.py:195
    d = F.conv2d(qn, protos[..., None, None]) * SIM_SCALE         # [1,P,h,w]
    return torch.sum(F.softmax(d, dim=1) * d, dim=1, keepdim=True)


def cls_unit(qry, sup_x, sup_y, mode, thresh, val_wsize):
    protos = get_prototypes(sup_x, sup_y, mode, val_wsize, thresh)
    if protos.shape[0] == 0:
        raise RuntimeError("failed to find prototypes")  # reference prints then F.conv2d raises (alpmodule.py:193-196)
    return predict(protos, qry, mode), protos


def fg_mode_for(fg_msk_hw, kernel_size):
    """grid_proto_fewshot.py:253-256: gridconv+ iff some kernel_size cell of the resized mask is >= 0.95."""
    return "gridconv+" if float(F.avg_pool2d(fg_msk_hw, kernel_size).max()) >= FG_THRESH else "mask"


def resize_to_patch_multiple(imgs, image_size, patch=14):
    s = image_size // patch * patch
    return F.interpolate(imgs, size=(s, s), mode="bilinear")      # grid_proto_fewshot.py:88-89


def features_to_map(tokens):
    """[B,HW,C] -> [B,C,h,w] (+ bilinear to 32x32 when fewer patches) (grid_proto_fewshot.py:91-98)."""
    B, HW, C = tokens.shape
    s = int(HW ** 0.5)
    fm = tokens.permute(0, 2, 1).reshape(B, C, s, s)
    if HW < DEFAULT_FEATURE_SIZE ** 2:
        fm = F.interpolate(fm, size=(DEFAULT_FEATURE_SIZE, DEFAULT_FEATURE_SIZE), mode="bilinear")
    return fm


def fewshot_scores(qry_ft, sup_ft, fg_mask, kernel_size, val_wsize=2, taps=None):
    """qry_ft/sup_ft [1,C,h,w]; fg_mask [1,H,W] in {0,1}. Returns pred [1,2,h,w] (bg, fg) before upsampling
    (grid_proto_fewshot.py:228-270)."""
    h, w = qry_ft.shape[-2:]
    fg = F.interpolate(fg_mask[None].float(), size=(h, w), mode="nearest")            # [1,1,h,w]
    bg = F.interpolate((1 - fg_mask)[None].float(), size=(h, w), mode="nearest")
    bg_score, bg_protos = cls_unit(qry_ft, sup_ft, bg, "gridconv", BG_THRESH, val_wsize)
    mode = fg_mode_for(fg, kernel_size)
    fg_score, fg_protos = cls_unit(qry_ft, sup_ft, fg, mode, FG_THRESH, val_wsize)
    if taps is not None:
        taps.update(bg_protos=bg_protos, fg_protos=fg_protos, fg_mode=mode, fg_msk=fg, bg_msk=bg)
    return torch.cat([bg_score, fg_score], dim=1)


def fewshot_scores_multishot(qry_ft, sup_fts, fg_masks, kernel_size, val_wsize=2):
    """n_shots > 1 (grid_proto_fewshot.py:228-262): sup_fts [n,C,h,w], fg_masks [n,H,W]. The background classifier sees ALL shots at
    once (their grid prototypes concatenated in shot order, alpmodule.py:111-131), the foreground one runs per shot - each with its
    own mode by the rule of :253-256 - and the shots' raw scores are combined by an element-wise max (:265-266)."""
    h, w = qry_ft.shape[-2:]
    fg = F.interpolate(fg_masks[:, None].float(), size=(h, w), mode="nearest")        # [n,1,h,w]
    bg = F.interpolate((1 - fg_masks)[:, None].float(), size=(h, w), mode="nearest")
    bg_score, _ = cls_unit(qry_ft, sup_fts, bg, "gridconv", BG_THRESH, val_wsize)
    raw = []
    for i in range(sup_fts.shape[0]):
        mode = fg_mode_for(fg[i:i + 1], kernel_size)
        raw.append(cls_unit(qry_ft, sup_fts[i:i + 1], fg[i:i + 1], mode, FG_THRESH, val_wsize)[0])
    fg_score = torch.stack(raw, dim=1).max(dim=1)[0]
    return torch.cat([bg_score, fg_score], dim=1)


def fewshot_forward_multishot(encode_fn, supp_imgs, fg_masks, qry_img, image_size, proto_grid_size=8, val_wsize=2):
    """FewShotSeg.forward with n_ways = 1, n_shots = len(supp_imgs) (grid_proto_fewshot.py:150-290). supp_imgs: list of [1,3,H,W],
    fg_masks: list of [1,H,W]."""
    n = len(supp_imgs)
    imgs = torch.cat(list(supp_imgs) + [qry_img], dim=0)                              # :181-182 shots first, the query last
    fm = features_to_map(encode_fn(resize_to_patch_multiple(imgs, image_size)))
    feature_hw = max(image_size // 14, DEFAULT_FEATURE_SIZE)
    ks = feature_hw // proto_grid_size
    pred = fewshot_scores_multishot(fm[n:n + 1], fm[:n], torch.cat(list(fg_masks), dim=0), ks, val_wsize)
    return F.interpolate(pred, size=supp_imgs[0].shape[-2:], mode="bilinear")


def fewshot_forward(encode_fn, supp_img, fg_mask, qry_img, image_size, proto_grid_size=8, val_wsize=2, taps=None):
    """FewShotSeg.forward for n_ways = n_shots = 1 (grid_proto_fewshot.py:150-290) with a DINOv2 encoder.
    encode_fn(imgs[B,3,s,s]) -> patch tokens [B, n, C].  Returns logits [1,2,H,W]."""
    imgs = torch.cat([supp_img, qry_img], dim=0)                                      # :181-182
    fm = features_to_map(encode_fn(resize_to_patch_multiple(imgs, image_size)))
    feature_hw = max(image_size // 14, DEFAULT_FEATURE_SIZE)                          # :59-60
    ks = feature_hw // proto_grid_size                                                # alpmodule.py:34
    if taps is not None:
        taps["img_fts"] = fm
    pred = fewshot_scores(fm[1:2], fm[0:1], fg_mask, ks, val_wsize, taps)
    if taps is not None:
        taps["pred_grid"] = pred
    return F.interpolate(pred, size=supp_img.shape[-2:], mode="bilinear")             # :272-273


def fewshot_forward_resnet(encode_map_fn, supp_img, fg_mask, qry_img, image_size, proto_grid_size=8, val_wsize=2, taps=None):
    """FewShotSeg.forward with `which_model = 'dlfcn_res101'` (grid_proto_fewshot.py:49-53,84-85): no resize to a patch
    multiple, features [B,256,ceil(S/8),ceil(S/8)] straight from the encoder. encode_map_fn(imgs[B,3,S,S]) -> [B,C,h,w]."""
    import math
    imgs = torch.cat([supp_img, qry_img], dim=0)
    fm = encode_map_fn(imgs)
    feature_hw = math.ceil(image_size / 8)
    assert fm.shape[-1] == feature_hw
    ks = feature_hw // proto_grid_size
    if taps is not None:
        taps["img_fts"] = fm
    pred = fewshot_scores(fm[1:2], fm[0:1], fg_mask, ks, val_wsize, taps)
    return F.interpolate(pred, size=supp_img.shape[-2:], mode="bilinear")
