"""Data contract either side of the 3-D path: the on-disk layout the reference's BTCV loader reads
(`func_3d/dataset/btcv.py:14-130`) and the validation flow that consumes it (`func_3d/function.py:196-330`), restated against the
drop-in `SAM2VideoPredictor` and the device-side `eval_seg`.

    <root>/{Training,Test}/image/<case>/<i>.jpg        RGB slices, i = 0 .. T-1
    <root>/{Training,Test}/mask/<case>/<i>.npy         integer label maps (0 = background, k = object k), same indices

The 2-D contract (`func_2d/dataset.py:16-100`, REFUGE optic-cup images with 7 raters):

    <root>/{Training,Test}-400/<name>/<name>_cropped.jpg                     RGB fundus crop
    <root>/{Training,Test}-400/<name>/<name>_seg_cup_<r>_cropped.jpg         rater r = 1 .. 7, grey level >= 0.5 = cup

`REFUGEImages[i]` returns the reference's dictionary (`image` [3,S,S] in 0..1, `multi_rater` [7,1,S,S], `p_label`, `pt` (row, col) as
`func_2d/utils.py:572-577` returns it, `mask` [1,out,out] majority vote resized, `mask_ori` [1,S,S], `image_meta_dict`).

`BTCVVolumes[i]` returns the same dictionary the reference's `BTCV.__getitem__` does (`image` [T,3,S,S] in 0..255, `label`
{frame: {obj: [1,S,S] int mask}}, `bbox` {frame: {obj: [4]}} or `pt` / `p_label`, `image_meta_dict`): leading / trailing slices
without any label are cropped, the volume is cut to `video_length` slices (default: a quarter of the labelled range, like the
reference's validation mode), images and masks are resized to `image_size` with PIL's default filters.  `write_synthetic_case`
produces a case in this layout from the synthetic blob generator (tests, smoke runs: there is no dataset in this environment).
"""
from __future__ import annotations

import os
import random
from typing import Dict, Optional

import numpy as np
import torch


def random_click(mask: np.ndarray, point_labels: int = 1, seed: Optional[int] = None):
    """func_3d/utils.py:89-105: a random pixel of the largest label value, as (label, [x, y])."""
    max_label = int(mask.max()) if mask.size else 0
    if max_label == 0:
        point_labels = max_label
    indices = np.argwhere(mask == max_label)
    rnd = random.Random(seed) if seed is not None else random
    r = rnd.randint(0, len(indices) - 1)
    return point_labels, np.array([indices[r][1], indices[r][0]])


def generate_bbox(mask: np.ndarray, variation: float = 0, seed: Optional[int] = None) -> np.ndarray:
    """func_3d/utils.py:107-137 (variation = 0 form): tight box of the largest label as [x0, y0, x1, y1] = (col, row) order;
    NaNs when the mask is empty."""
    if mask.ndim != 2:
        raise ValueError(f"Mask shape is not 2D, but {mask.shape}")
    max_label = mask.max() if mask.size else 0
    if max_label == 0:
        return np.array([np.nan] * 4)
    idx = np.argwhere(mask == max_label)
    r0, r1, c0, c1 = idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max()
    if variation > 0:
        rng = np.random.RandomState(seed)
        dw, dh = rng.randn(2) * variation
        mr, mc, h, w = (r0 + r1) / 2, (c0 + c1) / 2, (r1 - r0) * (1 + dw), (c1 - c0) * (1 + dh)
        r0, r1, c0, c1 = mr - h / 2, mr + h / 2, mc - w / 2, mc + w / 2
    return np.array([c0, r0, c1, r1])


class BTCVVolumes:
    """`func_3d/dataset/btcv.py:14-130` (a torch-style map dataset without the torch.utils.data dependency)."""

    def __init__(self, data_path: str, image_size: int = 1024, mode: str = "Test", prompt: str = "bbox", video_length: Optional[int] = None,
                 seed: Optional[int] = None, variation: float = 0):
        if prompt not in ("bbox", "click"):
            raise ValueError("Prompt not recognized")
        self.data_path, self.mode, self.prompt, self.img_size = data_path, mode, prompt, image_size
        self.video_length, self.seed, self.variation = video_length, seed, variation
        self.name_list = sorted(os.listdir(os.path.join(data_path, mode, "image")))

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, index: int) -> Dict:
        from PIL import Image
        name = self.name_list[index]
        img_path = os.path.join(self.data_path, self.mode, "image", name)
        mask_path = os.path.join(self.data_path, self.mode, "mask", name)
        n = len([f for f in os.listdir(mask_path) if f.endswith(".npy")])
        seg = np.stack([np.load(os.path.join(mask_path, f"{i}.npy")) for i in range(n)], axis=-1)
        labelled = [i for i in range(n) if seg[..., i].sum() > 0]
        first, last = (labelled[0], labelled[-1]) if labelled else (0, n - 1)
        seg = seg[..., first: last + 1]
        num_frame = seg.shape[-1]
        video_length = self.video_length if self.video_length is not None else max(1, int(num_frame / 4))
        video_length = min(video_length, num_frame)
        start = np.random.randint(0, num_frame - video_length + 1) if (num_frame > video_length and self.mode == "Training") else 0
        S = self.img_size
        img_tensor = torch.zeros(video_length, 3, S, S)
        mask_dict, bbox_dict, pt_dict, plabel_dict = {}, {}, {}, {}
        for fi in range(start, start + video_length):
            img = Image.open(os.path.join(img_path, f"{fi + first}.jpg")).convert("RGB").resize((S, S))
            img_tensor[fi - start] = torch.tensor(np.array(img)).permute(2, 0, 1)
            m = seg[..., fi]
            masks, boxes, pts, plabels = {}, {}, {}, {}
            for obj in np.unique(m[m > 0]):
                om = np.array(Image.fromarray(m == obj).resize((S, S)))
                masks[obj] = torch.tensor(om).unsqueeze(0).int()
                if self.prompt == "bbox":
                    boxes[obj] = torch.tensor(generate_bbox(om, variation=self.variation, seed=self.seed), dtype=torch.float32)
                else:
                    lab, xy = random_click(om, 1, seed=self.seed)
                    plabels[obj], pts[obj] = torch.tensor([lab], dtype=torch.int32), torch.tensor(xy[None], dtype=torch.float32)
            k = fi - start
            mask_dict[k] = masks
            if self.prompt == "bbox":
                bbox_dict[k] = boxes
            else:
                pt_dict[k], plabel_dict[k] = pts, plabels
        out = {"image": img_tensor, "label": mask_dict, "image_meta_dict": {"filename_or_obj": name}}
        if self.prompt == "bbox":
            out["bbox"] = bbox_dict
        else:
            out["pt"], out["p_label"] = pt_dict, plabel_dict
        return out


def random_click_2d(mask: np.ndarray, point_label: int = 1, rng: Optional[np.random.RandomState] = None):
    """func_2d/utils.py:572-577: a random position of the maximum value of a (soft) mask as (label, [row, col]); label 0 when the
    mask is empty."""
    max_label = mask.max() if mask.size else 0
    if round(float(max_label)) == 0:
        point_label = round(float(max_label))
    indices = np.argwhere(mask == max_label)
    r = (rng or np.random).randint(len(indices))
    return point_label, indices[r]


class REFUGEImages:
    """`func_2d/dataset.py:16-100` with the transform of train_2d.py:59-67 (Resize((S, S)) + ToTensor) built in."""
    N_RATERS = 7

    def __init__(self, data_path: str, image_size: int = 1024, out_size: int = 1024, mode: str = "Training", prompt: str = "click",
                 seed: Optional[int] = None):
        self.data_path, self.mode, self.prompt, self.img_size, self.mask_size = data_path, mode, prompt, image_size, out_size
        self.rng = np.random.RandomState(seed) if seed is not None else None
        base = os.path.join(data_path, mode + "-400")
        self.subfolders = sorted(os.path.join(base, f) for f in os.listdir(base) if os.path.isdir(os.path.join(base, f)))

    def __len__(self):
        return len(self.subfolders)

    def __getitem__(self, index: int) -> Dict:
        from PIL import Image
        sub = self.subfolders[index]
        name = os.path.basename(sub)
        S = self.img_size
        to_tensor = lambda im: torch.from_numpy(np.asarray(im.resize((S, S), Image.BILINEAR), dtype=np.float32) / 255.0)
        img = to_tensor(Image.open(os.path.join(sub, name + "_cropped.jpg")).convert("RGB")).permute(2, 0, 1).contiguous()
        raters = [(to_tensor(Image.open(os.path.join(sub, f"{name}_seg_cup_{r}_cropped.jpg")).convert("L")) >= 0.5).float()[None]
                  for r in range(1, self.N_RATERS + 1)]
        multi = torch.stack(raters, dim=0)                                      # [7, 1, S, S]
        out = {"image": img, "multi_rater": multi, "image_meta_dict": {"filename_or_obj": name}}
        if self.prompt == "click":
            mean = multi.mean(dim=0)                                             # [1, S, S] agreement in 0..1
            out["p_label"], out["pt"] = random_click_2d(mean.squeeze(0).numpy(), 1, self.rng)
            ori = (mean >= 0.5).float()
            small = torch.nn.functional.interpolate(ori.unsqueeze(0), size=(self.mask_size, self.mask_size), mode="bilinear",
                                                    align_corners=False).mean(dim=0)
            out["mask"], out["mask_ori"] = (small >= 0.5).float(), ori
        return out


def write_synthetic_refuge_case(root: str, name: str, size: int = 128, seed: int = 0, mode: str = "Training"):
    """One case in the REFUGE layout: a blob image and 7 rater masks that disagree at the rim (discs of slightly different radius)."""
    from PIL import Image
    from . import synthetic as syn
    rng = np.random.RandomState(seed)
    img, (cx, cy) = syn.blob_image(seed, size)                                # 0..255, centre of the brightest blob
    d = os.path.join(root, mode + "-400", name)
    os.makedirs(d, exist_ok=True)
    Image.fromarray(img.clamp(0, 255).permute(1, 2, 0).numpy().astype(np.uint8)).save(os.path.join(d, name + "_cropped.jpg"), quality=95)
    ys, xs = np.mgrid[0:size, 0:size]
    r0 = size / 8.0
    for r in range(1, REFUGEImages.N_RATERS + 1):
        rad = r0 * (0.85 + 0.3 * rng.rand())
        m = (((xs - cx) ** 2 + (ys - cy) ** 2) <= rad ** 2).astype(np.uint8) * 255
        Image.fromarray(m).save(os.path.join(d, f"{name}_seg_cup_{r}_cropped.jpg"), quality=95)


def write_synthetic_case(root: str, case: str, n_slices: int = 8, size: int = 128, n_objects: int = 2, seed: int = 0, mode: str = "Test"):
    """One case in the BTCV layout from the synthetic 3-D blob generator (labels from the blobs' iso-surfaces)."""
    from PIL import Image
    from . import synthetic as syn
    vol, boxes = syn.blob_volume(seed, n_slices, size, n_objects, normalised=False)
    idir, mdir = os.path.join(root, mode, "image", case), os.path.join(root, mode, "mask", case)
    os.makedirs(idir, exist_ok=True)
    os.makedirs(mdir, exist_ok=True)
    ys, xs = np.mgrid[0:size, 0:size]
    for t in range(n_slices):
        Image.fromarray(vol[t].clamp(0, 255).permute(1, 2, 0).numpy().astype(np.uint8)).save(os.path.join(idir, f"{t}.jpg"), quality=95)
        lab = np.zeros((size, size), dtype=np.int64)
        for o in range(n_objects):
            b = boxes[o][t]
            if b is not None:
                cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, max((b[2] - b[0]) / 2, 0.5), max((b[3] - b[1]) / 2, 0.5)
                lab[((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2 <= 1.0] = o + 1
        np.save(os.path.join(mdir, f"{t}.npy"), lab)


@torch.no_grad()
def validate_volume(net, pack: Dict, prompt: str = "bbox", prompt_freq: int = 2, threshold=(0.1, 0.3, 0.5, 0.7, 0.9)):
    """The per-volume body of `validation_sam` (func_3d/function.py:215-330) against the drop-in predictor: prompts on every
    `prompt_freq`-th slice (an empty mask where an object is absent), propagation from slice 0, mean IoU / Dice of `eval_seg` over
    all (slice, object) pairs.  Returns (iou, dice, video_segments)."""
    from .metrics import eval_seg
    dev = torch.device("cuda")
    imgs = pack["image"].to(dtype=torch.float32, device=dev)
    mask_dict = pack["label"]
    frame_id = list(range(imgs.size(0)))
    state = net.val_init_state(imgs_tensor=imgs)
    obj_list = sorted({o for f in frame_id for o in mask_dict[f].keys()})
    if not obj_list:
        return None
    for f in range(0, len(frame_id), prompt_freq):
        for obj in obj_list:
            try:
                if prompt == "click":
                    net.train_add_new_points(inference_state=state, frame_idx=f, obj_id=obj, points=pack["pt"][f][obj].to(dev),
                                             labels=pack["p_label"][f][obj].to(dev), clear_old_points=False)
                else:
                    net.train_add_new_bbox(inference_state=state, frame_idx=f, obj_id=obj, bbox=pack["bbox"][f][obj].to(dev),
                                           clear_old_points=False)
            except KeyError:
                net.train_add_new_mask(inference_state=state, frame_idx=f, obj_id=obj, mask=torch.zeros(imgs.shape[2:], device=dev))
    segments = {}
    for out_frame_idx, out_obj_ids, out_mask_logits in net.propagate_in_video(state, start_frame_idx=0):
        segments[out_frame_idx] = {o: out_mask_logits[i] for i, o in enumerate(out_obj_ids)}
    iou = dice = 0.0
    for f in frame_id:
        for obj in obj_list:
            pred = segments[f][obj].unsqueeze(0)
            gt = mask_dict[f].get(obj)
            gt = gt.to(dtype=torch.float32, device=dev).unsqueeze(0) if gt is not None else torch.zeros_like(pred)
            r = eval_seg(pred, gt, threshold)
            iou, dice = iou + r[0], dice + r[1]
    n = len(frame_id) * len(obj_list)
    net.reset_state(state)
    return iou / n, dice / n, segments
