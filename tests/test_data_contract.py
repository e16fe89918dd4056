"""Data contract (BTCV on-disk layout) and the validation flow on top of the drop-in predictor."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.data as D  # noqa: E402


def test_prompt_generators():
    m = np.zeros((10, 12), dtype=np.int32)
    m[2:5, 3:9] = 1
    assert D.generate_bbox(m).tolist() == [3, 2, 8, 4]                # [x0, y0, x1, y1]
    assert np.isnan(D.generate_bbox(np.zeros((4, 4), dtype=np.int32))).all()
    with pytest.raises(ValueError):
        D.generate_bbox(np.zeros((2, 2, 2)))
    lab, xy = D.random_click(m, 1, seed=3)
    assert lab == 1 and m[xy[1], xy[0]] == 1                          # (x, y) order
    lab, _ = D.random_click(np.zeros((4, 4), dtype=np.int32), 1, seed=3)
    assert lab == 0


def test_btcv_layout_roundtrip(tmp_path):
    root = str(tmp_path)
    D.write_synthetic_case(root, "case0", n_slices=10, size=64, n_objects=2, seed=1)
    D.write_synthetic_case(root, "case1", n_slices=6, size=64, n_objects=1, seed=2)
    assert sorted(os.listdir(os.path.join(root, "Test", "image"))) == ["case0", "case1"]
    ds = D.BTCVVolumes(root, image_size=32, mode="Test", prompt="bbox", video_length=None)
    assert len(ds) == 2
    s = ds[0]
    T = s["image"].shape[0]
    assert s["image"].shape[1:] == (3, 32, 32) and s["image"].max() <= 255 and s["image_meta_dict"]["filename_or_obj"] == "case0"
    assert set(s["label"].keys()) == set(range(T)) == set(s["bbox"].keys())
    # the first delivered slice carries a label (leading empty slices are cropped) and boxes bound their masks
    assert len(s["label"][0]) >= 1
    for f in range(T):
        for obj, msk in s["label"][f].items():
            assert msk.shape == (1, 32, 32) and msk.dtype == torch.int32
            x0, y0, x1, y1 = s["bbox"][f][obj].tolist()
            ys, xs = np.nonzero(msk[0].numpy())
            assert (xs.min(), ys.min(), xs.max(), ys.max()) == (x0, y0, x1, y1)
    full = D.BTCVVolumes(root, image_size=32, prompt="click", video_length=3)[1]
    n = full["image"].shape[0]
    assert 1 <= n <= 3 and set(full["pt"].keys()) == set(range(n))      # cut to video_length, never beyond the labelled range
    for f in range(n):
        for obj, xy in full["pt"][f].items():
            assert full["label"][f][obj][0, int(xy[0, 1]), int(xy[0, 0])] == 1 and full["p_label"][f][obj].tolist() == [1]
    with pytest.raises(ValueError):
        D.BTCVVolumes(root, prompt="scribble")


def test_refuge_layout_roundtrip(tmp_path):
    """func_2d/dataset.py:16-100: keys, shapes and value ranges of a REFUGE sample; the click lies where all raters agree; the
    majority-vote masks at both sizes."""
    root = str(tmp_path)
    for i in range(3):
        D.write_synthetic_refuge_case(root, f"g{i:04d}", size=96, seed=i)
    ds = D.REFUGEImages(root, image_size=64, out_size=32, mode="Training", prompt="click", seed=5)
    assert len(ds) == 3
    s = ds[1]
    assert set(s) == {"image", "multi_rater", "p_label", "pt", "mask", "mask_ori", "image_meta_dict"}
    assert s["image"].shape == (3, 64, 64) and 0.0 <= s["image"].min() and s["image"].max() <= 1.0
    assert s["multi_rater"].shape == (7, 1, 64, 64) and set(s["multi_rater"].unique().tolist()) <= {0.0, 1.0}
    assert s["mask_ori"].shape == (1, 64, 64) and s["mask"].shape == (1, 32, 32)
    assert s["image_meta_dict"]["filename_or_obj"] == "g0001" and s["p_label"] == 1
    mean = s["multi_rater"].mean(dim=0)[0]
    r, c = int(s["pt"][0]), int(s["pt"][1])                            # (row, col), func_2d/utils.py:572-577
    assert mean[r, c] == mean.max() and s["mask_ori"][0, r, c] == 1
    assert torch.equal(s["mask_ori"], (s["multi_rater"].mean(dim=0) >= 0.5).float())
    # raters disagree at the rim only: the vote is strictly between "all" and "any"
    any_, all_ = (mean > 0).sum().item(), (mean == 1).sum().item()
    assert all_ < s["mask_ori"].sum().item() < any_
    assert 0 < s["mask"].sum().item() < 32 * 32
    lab, _ = D.random_click_2d(np.zeros((4, 4), dtype=np.float32))
    assert lab == 0


@pytest.mark.gpu
def test_validation_flow_on_synthetic_case(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    root = str(tmp_path)
    D.write_synthetic_case(root, "vol", n_slices=12, size=128, n_objects=2, seed=4)
    net = bs.build_sam2_video_predictor("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    net.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    net = net.cuda().eval()
    for prompt in ("bbox", "click"):
        pack = D.BTCVVolumes(root, image_size=256, prompt=prompt, video_length=6, seed=5)[0]
        T = pack["image"].shape[0]
        assert T >= 3
        iou, dice, segments = D.validate_volume(net, pack, prompt=prompt, prompt_freq=2)
        assert sorted(segments) == list(range(T))
        assert all(v.shape == (1, 256, 256) for f in segments.values() for v in f.values())
        assert 0.0 <= iou <= 1.0 and 0.0 <= dice <= 1.0 + 1e-6
