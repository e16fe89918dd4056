"""Drop-in for `sam2_train/sam2_image_predictor.py:19-446` (single-image path): same constructor, `set_image`, `predict`,
`_prep_prompts`, `_predict`, `get_image_embedding`, `reset_predictor`, with the pre/post-processing of
`sam2_train/utils/transforms.py` (SAM2Transforms) done by device kernels instead of torchvision."""
from __future__ import annotations

import logging
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops


class SAM2ImagePredictor:
    def __init__(self, sam_model, mask_threshold=0.0, max_hole_area=0.0, max_sprinkle_area=0.0) -> None:
        self.model = sam_model
        self.mask_threshold = mask_threshold
        self.max_hole_area = max_hole_area
        self.max_sprinkle_area = max_sprinkle_area
        self.resolution = self.model.image_size
        self._is_image_set = False
        self._features = None
        self._orig_hw = None
        self._is_batch = False
        s = self.resolution
        self._bb_feat_sizes = [(s // 4, s // 4), (s // 8, s // 8), (s // 16, s // 16)]  # (256,128,64) at 1024 like the reference

    @property
    def device(self) -> torch.device:
        return self.model.device

    def reset_predictor(self) -> None:
        self._is_image_set = False
        self._features = None
        self._orig_hw = None
        self._is_batch = False

    # -- SAM2Transforms ------------------------------------------------------------------------------------------------
    def _transform_image(self, image: np.ndarray) -> torch.Tensor:
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        return ops.image_prep(img, self.resolution)

    def _transform_coords(self, coords: torch.Tensor, normalize: bool, orig_hw) -> torch.Tensor:
        """utils/transforms.py:45-62 (host-side scalar arithmetic on a handful of prompt coordinates)."""
        if normalize:
            assert orig_hw is not None
            h, w = orig_hw
            coords = coords.clone()
            coords[..., 0] = coords[..., 0] / w
            coords[..., 1] = coords[..., 1] / h
        return coords * self.resolution

    def _postprocess_masks(self, masks: torch.Tensor, orig_hw) -> torch.Tensor:
        """utils/transforms.py:74-98."""
        masks = masks.float().contiguous()
        B, C, h, w = masks.shape
        if self.max_hole_area > 0 or self.max_sprinkle_area > 0:
            flat = masks.reshape(B * C, 1, h, w)
            if flat.data_ptr() == masks.data_ptr():
                flat = flat.clone()
            if self.max_hole_area > 0:
                ops.fill_components_(flat, int(self.max_hole_area), self.mask_threshold, False, self.mask_threshold + 10.0)
            if self.max_sprinkle_area > 0:
                ops.fill_components_(flat, int(self.max_sprinkle_area), self.mask_threshold, True, self.mask_threshold - 10.0)
            masks = flat.reshape(B, C, h, w)
        return ops.bilinear_upsample(masks, int(orig_hw[0]), int(orig_hw[1]))

    # -- public API ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def set_image(self, image) -> None:
        self.reset_predictor()
        if isinstance(image, np.ndarray):
            logging.info("For numpy array image, we assume (HxWxC) format")
            self._orig_hw = [image.shape[:2]]
        else:  # PIL image
            w, h = image.size
            self._orig_hw = [(h, w)]
            image = np.asarray(image.convert("RGB"))
        input_image = self._transform_image(image)[None, ...]
        assert len(input_image.shape) == 4 and input_image.shape[1] == 3
        backbone_out = self.model.forward_image(input_image)
        _, vision_feats, _, _ = self.model._prepare_backbone_features(backbone_out)
        if self.model.directly_add_no_mem_embed:
            top = vision_feats[-1]
            L, B, C = top.shape
            y = ops.add_cast(top.transpose(0, 1), self.model.no_mem_embed.detach().float().expand(B, L, C), 1.0, torch.float32)
            vision_feats[-1] = y.view(B, L, C).transpose(0, 1)
        feats = [feat.permute(1, 2, 0).view(1, -1, *size) for feat, size in zip(vision_feats[::-1], self._bb_feat_sizes[::-1])][::-1]
        self._features = {"image_embed": feats[-1], "high_res_feats": feats[:-1]}
        self._is_image_set = True

    def predict(self, point_coords: Optional[np.ndarray] = None, point_labels: Optional[np.ndarray] = None,
                box: Optional[np.ndarray] = None, mask_input: Optional[np.ndarray] = None, multimask_output: bool = True,
                return_logits: bool = False, normalize_coords=True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        mask_input, unnorm_coords, labels, unnorm_box = self._prep_prompts(point_coords, point_labels, box, mask_input, normalize_coords)
        masks, iou_predictions, low_res_masks = self._predict(unnorm_coords, labels, unnorm_box, mask_input, multimask_output,
                                                              return_logits=return_logits)
        return (masks.squeeze(0).float().detach().cpu().numpy(), iou_predictions.squeeze(0).float().detach().cpu().numpy(),
                low_res_masks.squeeze(0).float().detach().cpu().numpy())

    def _prep_prompts(self, point_coords, point_labels, box, mask_logits, normalize_coords, img_idx=-1):
        unnorm_coords, labels, unnorm_box, mask_input = None, None, None, None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            point_coords = torch.as_tensor(point_coords, dtype=torch.float, device=self.device)
            unnorm_coords = self._transform_coords(point_coords, normalize_coords, self._orig_hw[img_idx])
            labels = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)
            if len(unnorm_coords.shape) == 2:
                unnorm_coords, labels = unnorm_coords[None, ...], labels[None, ...]
        if box is not None:
            box = torch.as_tensor(box, dtype=torch.float, device=self.device)
            unnorm_box = self._transform_coords(box.reshape(-1, 2, 2), normalize_coords, self._orig_hw[img_idx])
        if mask_logits is not None:
            mask_input = torch.as_tensor(mask_logits, dtype=torch.float, device=self.device)
            if len(mask_input.shape) == 3:
                mask_input = mask_input[None, :, :, :]
        return mask_input, unnorm_coords, labels, unnorm_box

    @torch.no_grad()
    def _predict(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output: bool = True,
                 return_logits: bool = False, img_idx: int = -1):
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        concat_points = (point_coords, point_labels) if point_coords is not None else None
        if boxes is not None:
            box_coords = boxes.reshape(-1, 2, 2)
            box_labels = torch.tensor([[2, 3]], dtype=torch.int, device=boxes.device).repeat(boxes.size(0), 1)
            if concat_points is not None:
                concat_points = (torch.cat([box_coords, concat_points[0]], dim=1), torch.cat([box_labels, concat_points[1]], dim=1))
            else:
                concat_points = (box_coords, box_labels)
        sparse_embeddings, dense_embeddings = self.model.sam_prompt_encoder(points=concat_points, boxes=None, masks=mask_input)
        batched_mode = concat_points is not None and concat_points[0].shape[0] > 1
        assert not batched_mode, "many prompt sets on one image (repeat_image) are outside this round's path"
        high_res_features = [lvl[img_idx].unsqueeze(0) for lvl in self._features["high_res_feats"]]
        low_res_masks, iou_predictions, _, _ = self.model.sam_mask_decoder(
            image_embeddings=self._features["image_embed"][img_idx].unsqueeze(0), image_pe=self.model.sam_prompt_encoder.get_dense_pe(),
            sparse_prompt_embeddings=sparse_embeddings, dense_prompt_embeddings=dense_embeddings, multimask_output=multimask_output,
            repeat_image=batched_mode, high_res_features=high_res_features)
        masks = self._postprocess_masks(low_res_masks, self._orig_hw[img_idx])
        low_res_masks = torch.clamp(low_res_masks, -32.0, 32.0)
        if not return_logits:
            masks = masks > self.mask_threshold
        return masks, iou_predictions, low_res_masks

    def get_image_embedding(self) -> torch.Tensor:
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        return self._features["image_embed"]
