"""The drop-in binding of the post-processor: ``bind_device_postprocess(RefVoxelPostprocessor)`` is a
subclass of the reference's class that overrides only post_process_airv2x, so the other methods the AirV2X dataset
calls on the same object (intermediate_fusion_dataset.py:360,482,806,935-936,960) stay the reference's."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.voxel_postprocessor import DevicePostprocess, VoxelPostprocessor, bind_device_postprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeReferencePostprocessor:
    """Shape of the reference's class: constructor fields (voxel_postprocessor.py:26-31) + the dataset-facing methods."""

    def __init__(self, anchor_params, dataset, train):
        self.params, self.dataset, self.train = anchor_params, dataset, train
        self.anchor_num = anchor_params["anchor_args"].get("num", 2)
        self.num_class = anchor_params["anchor_args"].get("num_class", 7)
        self.lidar_range = anchor_params["anchor_args"]["cav_lidar_range"]
        self.calls = []

    def generate_anchor_box(self):
        return VoxelPostprocessor(self.params).generate_anchor_box()

    def generate_label_airv2x(self, **kw):
        self.calls.append("generate_label_airv2x")
        return {"pos_equal_one": None}

    def generate_object_center_airv2x(self, cav_contents, pose):
        self.calls.append("generate_object_center_airv2x")
        return None

    @staticmethod
    def collate_batch_airv2x(batch):
        return {"collated": len(batch)}

    def generate_gt_bbx_airv2x(self, data_dict):
        self.calls.append("generate_gt_bbx_airv2x")
        return torch.zeros(0, 8, 3), [], []

    def post_process_segmentation_airv2x(self, data_dict, output_dict):
        self.calls.append("post_process_segmentation_airv2x")
        return None, None, None, None

    def post_process_airv2x(self, data_dict, output_dict):
        raise AssertionError("the reference's host post-process must be overridden")


def dataset_post_process(post_processor, data_dict, output_dict):
    """intermediate_fusion_dataset.py:932-938, verbatim call sequence."""
    pred_box_tensor, pred_score, pred_labels, pred_boxes3d = post_processor.post_process_airv2x(data_dict, output_dict)
    gt_box_tensor, gt_class_label_list, gt_track_list = post_processor.generate_gt_bbx_airv2x(data_dict)
    gt_box_tensor, gt_class_label_list, gt_track_list = post_processor.generate_gt_bbx_airv2x(data_dict)
    return pred_box_tensor, pred_score, pred_labels, pred_boxes3d, gt_box_tensor, gt_class_label_list, gt_track_list


def test_bound_class_keeps_every_other_reference_method():
    hy = synth.default_hypes()
    Bound = bind_device_postprocess(FakeReferencePostprocessor)
    assert issubclass(Bound, FakeReferencePostprocessor) and Bound.__name__ == "FakeReferencePostprocessor"
    pp = Bound(hy["postprocess"], "airv2x", False)
    assert Bound.post_process_airv2x is DevicePostprocess.post_process_airv2x
    for name in ("generate_label_airv2x", "generate_object_center_airv2x", "collate_batch_airv2x", "generate_gt_bbx_airv2x",
                 "post_process_segmentation_airv2x", "generate_anchor_box"):
        assert getattr(Bound, name) is getattr(FakeReferencePostprocessor, name), name
    assert pp.generate_label_airv2x() == {"pos_equal_one": None}
    assert Bound.collate_batch_airv2x([1, 2]) == {"collated": 2}
    assert pp.nms_top == 1000 and pp._ws == {}


@pytest.mark.gpu
def test_dataset_post_process_with_the_bound_class_on_the_gpu():
    """dataset.post_process's call sequence on the bound class: device boxes + the (fake) reference's GT path; the
    transformation matrix is a CUDA tensor as in the reference's inference flow (no host read of it)."""
    from tests.helpers import load_fixture
    fx = load_fixture("w2c_small_n1")
    hy = synth.default_hypes([float(v) for v in fx["lidar_range"]])
    anchors = VoxelPostprocessor(hy["postprocess"]).generate_anchor_box()
    data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(np.array(anchors))}}
    out = {"ego": {k: torch.from_numpy(fx[k]).cuda() for k in ("psm", "rm", "obj")}}
    Bound = bind_device_postprocess(FakeReferencePostprocessor)
    pp = Bound(hy["postprocess"], "airv2x", False)
    data["ego"]["transformation_matrix"] = data["ego"]["transformation_matrix"].to("cuda")
    res = dataset_post_process(pp, data, out)
    assert pp.calls == ["generate_gt_bbx_airv2x", "generate_gt_bbx_airv2x"]
    alone = VoxelPostprocessor(hy["postprocess"]).post_process_airv2x(
        {"ego": dict(data["ego"], transformation_matrix=data["ego"]["transformation_matrix"].cpu())}, out)
    for a, b in zip(res[:4], alone):
        assert torch.equal(a, b)
    assert res[0].shape[1:] == (8, 3) and res[4].shape == (0, 8, 3)


@pytest.mark.skipif(not os.path.exists("/root/reference/opencood"), reason="the reference tree only exists in the build container")
def test_binding_the_real_reference_class():
    """In the build container: bind the REAL reference class (imported with the third-party stubs of gen_golden.py) and
    run the reference's own generate_gt_bbx_airv2x + generate_anchor_box through the bound object."""
    code = r"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import gen_golden as g
os.chdir(tempfile.mkdtemp())
g.import_reference()
from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor as Ref
from airv2x_perception_amd.opencood_iface.voxel_postprocessor import DevicePostprocess, VoxelPostprocessor, bind_device_postprocess
hy = g.load_ref_hypes()
Bound = bind_device_postprocess(Ref)
pp = Bound(hy["postprocess"], dataset="airv2x", train=False)
assert isinstance(pp, Ref) and Bound.post_process_airv2x is DevicePostprocess.post_process_airv2x
for name in ("generate_label_airv2x", "collate_batch_airv2x", "generate_gt_bbx_airv2x", "post_process_segmentation_airv2x",
             "generate_object_center_airv2x", "generate_anchor_box", "delta_to_boxes3d"):
    assert getattr(Bound, name) is getattr(Ref, name), name
assert np.array_equal(pp.generate_anchor_box(), VoxelPostprocessor(hy["postprocess"]).generate_anchor_box())
centers = torch.zeros(1, 100, 7); centers[0, 0] = torch.tensor([10., 2., -1., 1.5, 1.6, 3.9, 0.3]); centers[0, 1] = torch.tensor([-30., -5., -1., 1.5, 1.6, 3.9, 1.0])
mask = torch.zeros(1, 100); mask[0, :2] = 1
data = {"ego": {"transformation_matrix": torch.eye(4), "object_bbx_center": centers, "object_bbx_mask": mask,
                "object_ids": [[7, 9]], "class_ids": [[1, 2]]}}
gt, cls, trk = pp.generate_gt_bbx_airv2x(data)
assert gt.shape == (2, 8, 3) and cls == [1, 2] and trk == [7, 9]
print("BOUND-OK")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BOUND-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
