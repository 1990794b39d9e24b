#!/usr/bin/env python3
"""Full-depth parity probe (GPU box): where does the error of sigmoid(low_res_masks) come from?

For each chosen slice of the benchmark's synthetic volume it runs the HIP pipeline and the CPU oracle and prints
  total      : HIP pipeline vs oracle pipeline                                  (what the north-star bounds by 1e-3)
  decoder    : HIP decoder vs oracle decoder, BOTH on the HIP image embedding   (the decoder's own arithmetic)
  encoder    : oracle decoder on the HIP embedding vs on the oracle embedding   (what the encoder's fp16 GEMMs cost)
plus the embedding error itself, the final-mask Dice and the score error.

  python tools/parity_probe.py --sam vit_h --slices 8,24,32,40,56 [--sam-depth N] [--fp16-decoder]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sam", default="vit_h")
    ap.add_argument("--sam-depth", type=int, default=None)
    ap.add_argument("--dino-depth", type=int, default=None)
    ap.add_argument("--slices", default="8,32,56")
    ap.add_argument("--kind", default="ct")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--fp16-decoder", action="store_true")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from oracle import alp as oalp, dinov2 as odino, glue, sam_prompt_decoder as odec
    from protosam_amd.metrics import dice
    from protosam_amd.runner import build_protosam, part_assign, run_slices, support_set
    from protosam_amd.synth import synth_volume
    torch.set_num_threads(min(os.cpu_count() or 1, args.threads))
    dev = torch.device("cuda:0")
    model, alp_sd = build_protosam(dev, sam_type=args.sam, image_size=512, seed=1234, sam_depth=args.sam_depth,
                                   dino_depth=args.dino_depth)
    model.sam.mask_decoder.image_side_fp16 = args.fp16_decoder
    vol, lab = synth_volume(args.n, 512, seed=0, kind=args.kind)
    svol, slab = synth_volume(args.n, 512, seed=1, kind=args.kind)
    sup_imgs, sup_masks = support_set(svol, slab)
    sup_d, msk_d = [s.to(dev) for s in sup_imgs], [m.to(dev) for m in sup_masks]
    enc_sd = {k[len("encoder."):]: v for k, v in alp_sd.items() if k.startswith("encoder.")}
    sam_sd = {k: v.detach().cpu().float() for k, v in model.sam.state_dict().items()}
    enc = lambda im: odino.forward_features(im, enc_sd, "dinov2_b14", depth=args.dino_depth)["x_norm_patchtokens"]  # noqa
    captured = {}
    ie = model.sam.image_encoder
    orig = ie.encode_patches

    def cap(patches, B, **kw):
        out = orig(patches, B, **kw)
        captured["feat"] = out.clone()
        captured["x"] = ie._ws[B]["x"].clone()          # residual stream after the last block, token-major [B*4096, D]
        return out

    def oracle_neck(x_tok):
        """image_encoder.py:120 on the HIP residual stream: separates the blocks' error from the neck's."""
        import torch.nn.functional as F
        from oracle import sam_image_encoder as oenc
        pre = "image_encoder."
        x = x_tok.reshape(1, 64, 64, -1).permute(0, 3, 1, 2)
        x = F.conv2d(x, sam_sd[pre + "neck.0.weight"])
        x = oenc.layer_norm_2d(x, sam_sd[pre + "neck.1.weight"], sam_sd[pre + "neck.1.bias"])
        x = F.conv2d(x, sam_sd[pre + "neck.2.weight"], padding=1)
        return oenc.layer_norm_2d(x, sam_sd[pre + "neck.3.weight"], sam_sd[pre + "neck.3.bias"])
    ie.encode_patches = cap
    rows = []
    for z in [int(v) for v in args.slices.split(",")]:
        masks, _ = run_slices(model, vol.to(dev), sup_d, msk_d, [z], dev)
        st = model.last_stats
        g = masks[0].cpu().float()
        if "low_res" not in st:
            print(f"z={z}: empty coarse mask on the GPU path")
            continue
        low = st["low_res"][:, st["sel"]].cpu()
        iou = st["iou"][:, st["sel"]].cpu().numpy()
        feat_hip = captured["feat"][0].cpu().reshape(64, 64, 256).permute(2, 0, 1)[None].contiguous()
        q = vol[z][None, None].repeat(1, 3, 1, 1).contiguous()
        part = part_assign(z, args.n)
        t0 = time.time()
        with torch.no_grad():
            logits = oalp.fewshot_forward(enc, sup_imgs[part], sup_masks[part], q, 512)
            taps = {}
            pred_ref, scores_ref = glue.protosam_forward(q, logits, sam_sd, args.sam, use_bbox=True, use_points=True,
                                                         point_mode="both", use_cca=False, encoder_depth=args.sam_depth,
                                                         taps=taps)
            taps2 = {}
            glue.protosam_forward(q, logits, sam_sd, args.sam, use_bbox=True, use_points=True, point_mode="both",
                                  use_cca=False, encoder_depth=args.sam_depth, taps=taps2, features=feat_hip)
            taps3 = {}
            feat_blocks = oracle_neck(captured["x"][:4096].cpu().float())
            glue.protosam_forward(q, logits, sam_sd, args.sam, use_bbox=True, use_points=True, point_mode="both",
                                  use_cca=False, encoder_depth=args.sam_depth, taps=taps3, features=feat_blocks)
        dt = time.time() - t0
        if len(taps["low_res"]) != low.shape[0]:
            print(f"z={z}: component count differs ({low.shape[0]} vs {len(taps['low_res'])})")
            continue
        low_ref = torch.stack([l[0] for l in taps["low_res"]])
        low_mix = torch.stack([l[0] for l in taps2["low_res"]])
        sg = torch.sigmoid
        e_tot = (sg(low) - sg(low_ref)).abs().max().item()
        e_dec = (sg(low) - sg(low_mix)).abs().max().item()
        e_enc = (sg(low_mix) - sg(low_ref)).abs().max().item()
        low_blk = torch.stack([l[0] for l in taps3["low_res"]])
        e_blocks = (sg(low_blk) - sg(low_ref)).abs().max().item()      # error of the 12 / 32 blocks alone (exact neck)
        e_neck = (sg(low_mix) - sg(low_blk)).abs().max().item()        # what the fp16 neck adds on top
        fe = (feat_hip - taps["features"]).abs()
        row = dict(z=z, comps=int(low.shape[0]), total=e_tot, decoder=e_dec, encoder=e_enc, blocks=e_blocks, neck=e_neck, feat_max=fe.max().item(),
                   feat_mean=fe.mean().item(), dice=dice(g, pred_ref), flips=int((g != pred_ref.float()).sum()),
                   dscore=float(np.abs(iou - np.array(scores_ref)).max()), logit_max=low_ref.abs().max().item(),
                   oracle_s=round(dt, 1))
        rows.append(row)
        print(json.dumps(row), flush=True)
    if rows:
        print("worst total %.3e decoder %.3e encoder %.3e" % (max(r["total"] for r in rows), max(r["decoder"] for r in rows),
                                                              max(r["encoder"] for r in rows)))
    if args.out:
        json.dump(dict(args=vars(args), rows=rows), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
