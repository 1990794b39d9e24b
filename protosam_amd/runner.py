"""Volume-level driver: the per-slice loop of /root/reference/validation_protosam.py:346-388 (support per z-part,
one `ProtoSAM.forward` per query slice) plus data-parallel sharding of slices over ranks.

Slices are independent given the (replicated) support set, so rank r of W takes slices z = r (mod W) with no
data-path collective; the only exchange is one all-gather of the uint8 masks per volume / step.
"""
import math

import torch
import torch.distributed as dist

from .grid_proto_fewshot import FewShotSeg
from .protosam import ALPNetWrapper, InputFactory, ProtoSAM, TYPE_ALPNET
from .synth import synth_state_dict

ALP_CFG = {"which_model": "dinov2_b14", "cls_name": "grid_proto", "proto_grid_size": 8, "lora": 0, "align": False,
           "debug": False}


def build_protosam(device, sam_type="vit_h", image_size=512, seed=1234, dino_depth=None, sam_depth=None,
                   cache_support=True, heavy_tail=False, **protosam_kw):
    """Seeded synthetic-weight ProtoSAM as validation_protosam.get_model builds it (:188-232)."""
    cfg = dict(ALP_CFG)
    if dino_depth is not None:
        cfg["encoder_depth"] = dino_depth
    alp = FewShotSeg(image_size, None, cfg, cache_support=cache_support)
    alp_sd = synth_state_dict(alp, seed)
    alp.load_state_dict(alp_sd)
    alp = alp.to(device).eval()
    spec = ("random-heavy" if heavy_tail else "random") + f":{sam_type}:{seed}" + (f":{sam_depth}" if sam_depth is not None else "")
    kw = dict(use_bbox=True, use_points=True, point_mode="both", use_cca=False, num_points_for_sam=1,
              use_sam_trans=True)
    kw.update(protosam_kw)
    model = ProtoSAM(image_size=(1024, 1024), coarse_segmentation_model=ALPNetWrapper(alp), sam_pretrained_path=spec,
                     **kw).to(device).eval()
    return model, alp_sd


def part_assign(z, n_slices, n_parts=3):
    """dataloaders/common.py:241-249 style: equal z-chunks."""
    return min(int(z * n_parts / n_slices), n_parts - 1)


def support_set(vol, lab, n_parts=3):
    """Middle slice of each z-chunk of the support volume (ManualAnnoDatasetv2.py:457-462), tiled x3."""
    n = vol.shape[0]
    imgs, masks = [], []
    for p in range(n_parts):
        z = int((p + 0.5) * n / n_parts)
        imgs.append(vol[z][None, None].repeat(1, 3, 1, 1).contiguous())
        masks.append(lab[z][None].contiguous())
    return imgs, masks


def shard_slices(n_slices, rank, world):
    return list(range(rank, n_slices, world))


@torch.no_grad()
def run_slices(model, vol, sup_imgs, sup_masks, zs, device, out=None, batch=1, mix_parts=True):
    """Runs ProtoSAM on slices `zs` of `vol` [n,S,S] (device tensor), `batch` slices at a time through `forward_batch`; returns uint8
    masks [len(zs),S,S] and the number of prompts per slice. A batch may span z-parts (different support sets: the batch then carries
    one (input, count) pair per part - only the prototype match is done per part); mix_parts=False cuts batches at part boundaries."""
    n, S = vol.shape[0], vol.shape[-1]
    if out is None:
        out = torch.zeros((len(zs), S, S), dtype=torch.uint8, device=device)
    stats = [0] * len(zs)
    inputs = {}

    def part_input(part, q):
        if part not in inputs:
            inputs[part] = InputFactory.create_input(TYPE_ALPNET, q, support_images=[sup_imgs[part]],
                                                     support_labels=[sup_masks[part]], isval=True, val_wsize=2)
        return inputs[part]
    i = 0
    while i < len(zs):
        part = part_assign(zs[i], n)
        j = i
        while j < len(zs) and j - i < batch and (mix_parts and batch > 1 or part_assign(zs[j], n) == part):
            j += 1
        idx = torch.tensor(zs[i:j], device=device)
        q = vol[idx][:, None].expand(j - i, 3, S, S).contiguous()
        if batch == 1:
            res = [model(q, part_input(part, q), degrees_rotate=0)]
            st = [model.last_stats]
        else:
            runs = []                                   # consecutive slices of one part
            for z in zs[i:j]:
                pz = part_assign(z, n)
                if runs and runs[-1][0] == pz:
                    runs[-1][1] += 1
                else:
                    runs.append([pz, 1])
            cin = part_input(part, q) if len(runs) == 1 else [(part_input(pz, q), cnt) for pz, cnt in runs]
            res = model.forward_batch(q, cin)
            st = model.last_stats.get("per_slice", [model.last_stats])
        for k, (pred, scores) in enumerate(res):
            if pred.shape[-1] == S:
                out[i + k].copy_(pred)            # ({0., 1.} -> uint8 inside the one copy kernel)
            else:   # empty coarse mask: the reference hands back the all-zero 1024x1024 arg-max map (ProtoSAM.py:612-613)
                out[i + k].zero_()
            stats[i + k] = st[k].get("n_prompts", 0) if k < len(st) else 0
        i = j
    return out, stats


def gather_masks(local, world):
    """One all-gather of the per-rank uint8 masks (RCCL on GPUs, gloo on CPU). [k,S,S] -> [world*k,S,S] (rank-major)."""
    if world == 1 or not dist.is_initialized():
        return local
    return all_gather_rows(local)


def all_gather_rows(local):
    """dist.all_gather_into_tensor along dim 0. Device tensors go through RCCL as they are; under the gloo backend (the CPU tests, and
    the two-ranks-on-one-GPU test of the real pipeline, where RCCL refuses two ranks on one device) they are staged through the host."""
    world = dist.get_world_size()
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "gloo":
        host = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu())
        return host.to(local.device)
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local)
    return full


def interleave_rank_major(full, n_slices, world):
    """Undo the z = r (mod W) sharding: rank-major gathered masks -> z order (pads dropped)."""
    k = math.ceil(n_slices / world)
    idx = []
    for z in range(n_slices):
        idx.append((z % world) * k + z // world)
    return full[torch.tensor(idx, device=full.device)]
