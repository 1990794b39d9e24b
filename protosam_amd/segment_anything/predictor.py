"""`SamPredictor` (models/segment_anything/predictor.py:17-269) over the HIP-backed `Sam`.

Same methods, arguments, return types (numpy from `predict`, tensors from `predict_torch`) and the same RuntimeError
when predicting before `set_image`. ProtoSAM hands over 1024x1024 images (models/ProtoSAM.py:592-594,651-660), for
which `apply_image`, the padding and the second post-processing resize are identities; other sizes take the reference's
route: PIL resize on the host, zero padding after normalisation, crop + resize of the mask logits.
"""
import numpy as np
import torch

from .utils.transforms import ResizeLongestSide


class SamPredictor:
    def __init__(self, sam_model):
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.image_encoder.img_size)
        self.reset_image()

    def set_image(self, image, image_format="RGB"):
        assert image_format in ["RGB", "BGR"], f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        input_image = self.transform.apply_image(image)
        t = torch.as_tensor(np.ascontiguousarray(input_image), device=self.device)
        t = t.permute(2, 0, 1).contiguous()[None, :, :, :]
        self.set_torch_image(t, image.shape[:2])

    @torch.no_grad()
    def set_torch_image(self, transformed_image, original_image_size):
        S = self.model.image_encoder.img_size
        assert (len(transformed_image.shape) == 4 and transformed_image.shape[1] == 3
                and max(*transformed_image.shape[2:]) == S), f"set_torch_image input must be BCHW with long side {S}."
        self.reset_image()
        self.original_size = original_image_size
        self.input_size = tuple(transformed_image.shape[-2:])
        input_image = self.model.preprocess(transformed_image)
        self.features_tokens = self.model.image_encoder.forward_tokens(input_image).clone()  # [1, 4096, 256]
        g = self.model.image_encoder.grid
        self.features = self.features_tokens.view(1, g, g, -1).permute(0, 3, 1, 2)
        self.is_image_set = True

    def set_features_tokens(self, tokens, original_image_size, input_size):
        """Fast-path hook: adopt an embedding computed elsewhere (ProtoSAM's fused quantise+patchify+encode)."""
        self.reset_image()
        self.original_size, self.input_size = original_image_size, tuple(input_size)
        self.features_tokens = tokens
        g = self.model.image_encoder.grid
        self.features = tokens.view(1, g, g, -1).permute(0, 3, 1, 2)
        self.is_image_set = True

    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True,
                return_logits=False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        coords_torch, labels_torch, box_torch, mask_input_torch = None, None, None, None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            point_coords = self.transform.apply_coords(point_coords, self.original_size)
            coords_torch = torch.as_tensor(point_coords, dtype=torch.float, device=self.device)
            labels_torch = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)
            coords_torch, labels_torch = coords_torch[None, :, :], labels_torch[None, :]
        if box is not None:
            box = self.transform.apply_boxes(box, self.original_size)
            box_torch = torch.as_tensor(box, dtype=torch.float, device=self.device)
            box_torch = box_torch[None, :] if box_torch.dim() == 1 else box_torch
        if mask_input is not None:
            mask_input_torch = torch.as_tensor(mask_input, dtype=torch.float, device=self.device)[None, :, :, :]
        masks, iou_predictions, low_res_masks = self.predict_torch(coords_torch, labels_torch, box_torch,
                                                                   mask_input_torch, multimask_output,
                                                                   return_logits=return_logits)
        return (masks[0].detach().cpu().numpy(), iou_predictions[0].detach().cpu().numpy(),
                low_res_masks[0].detach().cpu().numpy())

    @torch.no_grad()
    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output=True,
                      return_logits=False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        points = (point_coords, point_labels) if point_coords is not None else None
        sparse_embeddings, dense_embeddings = self.model.prompt_encoder(points=points, boxes=boxes, masks=mask_input)
        low_res_masks, iou_predictions = self.model.mask_decoder(
            image_embeddings=self.features, image_pe=self.model.prompt_encoder.get_dense_pe(),
            sparse_prompt_embeddings=sparse_embeddings, dense_prompt_embeddings=dense_embeddings,
            multimask_output=multimask_output)
        low_res_masks = low_res_masks.contiguous()
        masks = self.model.postprocess_masks(low_res_masks, self.input_size, self.original_size)
        if not return_logits:
            masks = masks > self.model.mask_threshold
        return masks, iou_predictions, low_res_masks

    def get_image_embedding(self):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        assert self.features is not None, "Features must exist if an image has been set."
        return self.features

    @property
    def device(self):
        return self.model.device

    def reset_image(self):
        self.is_image_set = False
        self.features = None
        self.features_tokens = None
        self.orig_h = None
        self.orig_w = None
        self.input_h = None
        self.input_w = None
