"""CPU-side checks of the predictor mirrors: API surface (names / signatures of sam2_video_predictor.py and build_sam.py) and the
pure host logic of the video state machine that needs no kernel (object bookkeeping, reset, frame normalisation)."""
import inspect
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402
from medical_sam2_amd.video_predictor import SAM2VideoPredictor, concat_points, load_video_frames_from_data  # noqa: E402

# public + semi-public methods func_3d/function.py and the upstream notebooks call (sam2_video_predictor.py:39-1441)
REFERENCE_METHODS = {
    "init_state": ["video_path", "offload_video_to_cpu", "offload_state_to_cpu", "async_loading_frames"],
    "val_init_state": ["imgs_tensor", "video_height", "video_width", "offload_video_to_cpu", "offload_state_to_cpu", "async_loading_frames"],
    "train_init_state": ["imgs_tensor", "video_height", "video_width", "offload_video_to_cpu", "offload_state_to_cpu", "async_loading_frames"],
    "add_new_points": ["inference_state", "frame_idx", "obj_id", "points", "labels", "clear_old_points", "normalize_coords"],
    "train_add_new_points": ["inference_state", "frame_idx", "obj_id", "points", "labels", "clear_old_points", "normalize_coords"],
    "add_new_bbox": ["inference_state", "frame_idx", "obj_id", "bbox", "clear_old_points", "normalize_coords"],
    "train_add_new_bbox": ["inference_state", "frame_idx", "obj_id", "bbox", "clear_old_points", "normalize_coords"],
    "add_new_mask": ["inference_state", "frame_idx", "obj_id", "mask"],
    "train_add_new_mask": ["inference_state", "frame_idx", "obj_id", "mask"],
    "propagate_in_video_preflight": ["inference_state"],
    "propagate_in_video": ["inference_state", "start_frame_idx", "max_frame_num_to_track", "reverse"],
    "train_propagate_in_video": ["inference_state", "start_frame_idx", "max_frame_num_to_track", "reverse"],
    "reset_state": ["inference_state"],
    "_get_image_feature": ["inference_state", "frame_idx", "batch_size"],
    "_run_single_frame_inference": ["inference_state", "output_dict", "frame_idx", "batch_size", "is_init_cond_frame", "point_inputs",
                                    "mask_inputs", "reverse", "run_mem_encoder", "prev_sam_mask_logits"],
    "_run_memory_encoder": ["inference_state", "frame_idx", "batch_size", "high_res_masks", "is_mask_from_pts"],
    "_consolidate_temp_output_across_obj": ["inference_state", "frame_idx", "is_cond", "run_mem_encoder", "consolidate_at_video_res"],
    "_get_orig_video_res_output": ["inference_state", "any_res_masks"],
    "_add_output_per_object": ["inference_state", "frame_idx", "current_out", "storage_key"],
    "_clear_non_cond_mem_around_input": ["inference_state", "frame_idx"],
    "_get_maskmem_pos_enc": ["inference_state", "current_out"],
    "_get_empty_mask_ptr": ["inference_state", "frame_idx"],
    "_obj_id_to_idx": ["inference_state", "obj_id"],
    "_obj_idx_to_id": ["inference_state", "obj_idx"],
    "_get_obj_num": ["inference_state"],
}


def test_video_predictor_surface():
    for name, params in REFERENCE_METHODS.items():
        fn = getattr(SAM2VideoPredictor, name)
        got = [p for p in inspect.signature(fn).parameters if p != "self"]
        assert got == params, (name, got)


@pytest.fixture(scope="module")
def predictor():
    m = bs.build_sam2_video_predictor("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)      # same keys as SAM2Base: checkpoints load unchanged
    return m


def test_builder_overrides(predictor):
    assert isinstance(predictor, SAM2VideoPredictor)
    assert predictor.fill_hole_area == 8 and predictor.binarize_mask_from_pts_for_mem_enc and not predictor.non_overlap_masks
    assert predictor.sam_mask_decoder.dynamic_multimask_via_stability
    plain = bs.build_sam2_video_predictor("sam2_hiera_t", device="cpu", apply_postprocessing=False)
    assert plain.fill_hole_area == 0 and not plain.binarize_mask_from_pts_for_mem_enc


def _bare_state(predictor, n_frames=4):
    """the bookkeeping part of an inference state (what _new_state builds before it touches the GPU)"""
    from collections import OrderedDict
    return {"images": torch.zeros(n_frames, 3, 8, 8), "num_frames": n_frames, "video_height": 8, "video_width": 8,
            "device": torch.device("cpu"), "storage_device": torch.device("cpu"), "point_inputs_per_obj": {}, "mask_inputs_per_obj": {},
            "cached_features": {}, "constants": {}, "obj_id_to_idx": OrderedDict(), "obj_idx_to_id": OrderedDict(), "obj_ids": [],
            "output_dict": {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}, "output_dict_per_obj": {},
            "temp_output_dict_per_obj": {}, "consolidated_frame_inds": {"cond_frame_outputs": set(), "non_cond_frame_outputs": set()},
            "tracking_has_started": False, "frames_already_tracked": {}}


def test_object_bookkeeping_and_reset(predictor):
    st = _bare_state(predictor)
    assert predictor._obj_id_to_idx(st, 42) == 0 and predictor._obj_id_to_idx(st, 7) == 1 and predictor._obj_id_to_idx(st, 42) == 0
    assert st["obj_ids"] == [42, 7] and predictor._obj_idx_to_id(st, 1) == 7 and predictor._get_obj_num(st) == 2
    st["tracking_has_started"] = True
    with pytest.raises(RuntimeError, match="Cannot add new object id 5 after tracking starts"):
        predictor._obj_id_to_idx(st, 5)
    # split a 2-object output into per-object views that share storage, then clear around an input frame
    out = {"maskmem_features": torch.zeros(2, 64, 4, 4), "maskmem_pos_enc": [torch.zeros(2, 64, 4, 4)], "pred_masks": torch.zeros(2, 1, 16, 16),
           "obj_ptr": torch.zeros(2, 256)}
    for t in range(4):
        st["output_dict"]["non_cond_frame_outputs"][t] = out
        predictor._add_output_per_object(st, t, out, "non_cond_frame_outputs")
    view = st["output_dict_per_obj"][1]["non_cond_frame_outputs"][2]
    assert view["pred_masks"].shape == (1, 1, 16, 16) and view["pred_masks"].data_ptr() == out["pred_masks"][1:].data_ptr()
    predictor._clear_non_cond_mem_around_input(st, 100)
    assert len(st["output_dict"]["non_cond_frame_outputs"]) == 4
    predictor._clear_non_cond_mem_around_input(st, 2)
    assert not st["output_dict"]["non_cond_frame_outputs"] and not st["output_dict_per_obj"][0]["non_cond_frame_outputs"]
    predictor.reset_state(st)
    assert st["obj_ids"] == [] and not st["tracking_has_started"] and not st["output_dict_per_obj"]
    # constant memory position encoding: one copy per session, expanded per call
    pe = [torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).reshape(2, 3, 2, 2)]
    a = predictor._get_maskmem_pos_enc(st, {"maskmem_pos_enc": pe})
    b = predictor._get_maskmem_pos_enc(st, {"maskmem_pos_enc": [torch.zeros(5, 3, 2, 2)]})
    assert a[0].shape == (2, 3, 2, 2) and b[0].shape == (5, 3, 2, 2) and torch.equal(b[0][4], pe[0][0])
    assert predictor._get_maskmem_pos_enc(st, {"maskmem_pos_enc": None}) is None


def test_frame_normalisation_and_points():
    x = torch.full((2, 3, 4, 4), 255.0)
    y = load_video_frames_from_data(x, offload_video_to_cpu=True)
    ref = (1.0 - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])
    assert torch.allclose(y[0, :, 0, 0], ref, atol=1e-6)
    p = concat_points(None, torch.zeros(1, 1, 2), torch.ones(1, 1, dtype=torch.int32))
    p = concat_points(p, torch.ones(1, 2, 2), torch.zeros(1, 2, dtype=torch.int32))
    assert p["point_coords"].shape == (1, 3, 2) and p["point_labels"].tolist() == [[1, 0, 0]]
