"""SAM prompt encoder (names of models/segment_anything/modeling/prompt_encoder.py:16-214).

Points / boxes -> sparse tokens and the dense positional grid are produced by csrc/decoder.hip
(`psam_prompt_tokens`, `psam_dense_pe`); mask prompts (`mask_downscaling`, :51-59,102-105) by `psam_mask_downscale`
(the three convolutions, two LayerNorm2d and GELUs in one kernel).
"""
import torch
import torch.nn as nn

from ... import ops
from .common import LayerNorm2d, f32


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats=64, scale=None):
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim, image_embedding_size, input_image_size, mask_in_chans, activation=nn.GELU):
        super().__init__()
        assert embed_dim == 256
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans // 4), activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans),
            activation(), nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1))
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self._cache = None

    def _apply(self, fn, *a, **k):
        self._cache = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load
        self._cache = None
        return super()._load_from_state_dict(*a, **k)

    def _packed(self):
        if self._cache is None:
            G = f32(self.pe_layer.positional_encoding_gaussian_matrix)
            type_emb = torch.cat([f32(self.not_a_point_embed.weight)] + [f32(e.weight) for e in self.point_embeddings], 0)
            gh, gw = self.image_embedding_size
            md = self.mask_downscaling
            mask_w = torch.cat([f32(md[0].weight).reshape(-1), f32(md[0].bias), f32(md[1].weight), f32(md[1].bias),
                                f32(md[3].weight).reshape(-1), f32(md[3].bias), f32(md[4].weight), f32(md[4].bias),
                                f32(md[6].weight).reshape(-1), f32(md[6].bias)]).contiguous()
            self._cache = dict(G=G, type_emb=type_emb.contiguous(), pe_tok=ops.dense_pe(G, gh, gw),
                               no_mask=f32(self.no_mask_embed.weight).reshape(-1), mask_w=mask_w,
                               mask_eps=float(md[1].eps))
        return self._cache

    def get_dense_pe_tokens(self):
        """token-major [gh*gw, 256] (input independent, computed once)."""
        return self._packed()["pe_tok"]

    def get_dense_pe(self):
        gh, gw = self.image_embedding_size
        return self.get_dense_pe_tokens().view(gh, gw, self.embed_dim).permute(2, 0, 1).unsqueeze(0)

    def prompt_arrays(self, points, boxes):
        """-> coords fp32 [B,Ns,2], labels int32 [B,Ns] in the kernel's convention (corner labels 2/3, pad -1)."""
        cs, ls = [], []
        if points is not None:
            coords, labels = points
            coords, labels = coords.float(), labels.to(torch.int32)
            if boxes is None:  # prompt_encoder.py:80-84,155
                coords = torch.cat([coords, torch.zeros((coords.shape[0], 1, 2), device=coords.device)], dim=1)
                labels = torch.cat([labels, -torch.ones((labels.shape[0], 1), dtype=torch.int32, device=labels.device)], 1)
            cs.append(coords)
            ls.append(labels)
        if boxes is not None:
            b = boxes.float().reshape(-1, 2, 2)
            cs.append(b)
            ls.append(torch.tensor([2, 3], dtype=torch.int32, device=b.device).expand(b.shape[0], 2))
        if not cs:
            return None, None
        return torch.cat(cs, dim=1).contiguous(), torch.cat(ls, dim=1).contiguous()

    def embed_masks_tokens(self, masks):
        """masks [n,1,4g,4g] (any real dtype) -> token-major dense embeddings fp32 [n, g*g, 256]."""
        pk = self._packed()
        gh, gw = self.image_embedding_size
        if gh != gw or tuple(masks.shape[-2:]) != self.mask_input_size:
            raise ValueError(f"mask prompts must be {self.mask_input_size}, got {tuple(masks.shape[-2:])}")
        m = masks.to(device=pk["G"].device, dtype=torch.float32).reshape(-1, 4 * gh, 4 * gw).contiguous()
        return ops.mask_downscale(m, pk["mask_w"], gh, pk["mask_eps"])

    def forward(self, points, boxes, masks):
        pk = self._packed()
        dev = pk["G"].device
        coords, labels = self.prompt_arrays(points, boxes)
        bs = masks.shape[0] if (coords is None and masks is not None) else (1 if coords is None else coords.shape[0])
        if coords is None:
            sparse = torch.empty((bs, 0, self.embed_dim), device=dev)
        else:
            Ns = coords.shape[1]
            dummy = torch.zeros((5, 256), dtype=torch.float32, device=dev)
            tok = ops.prompt_tokens(coords.to(dev), labels.to(dev), pk["G"], pk["type_emb"], dummy, bs, Ns,
                                    self.input_image_size[0])
            sparse = tok[:, 5:]
        gh, gw = self.image_embedding_size
        if masks is not None:                                                    # prompt_encoder.py:163-164
            dense = self.embed_masks_tokens(masks).view(-1, gh, gw, self.embed_dim).permute(0, 3, 1, 2)
        else:
            dense = pk["no_mask"].reshape(1, -1, 1, 1).expand(bs, -1, gh, gw)
        return sparse, dense
