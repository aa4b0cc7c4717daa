"""Checkpoint loader: both layouts the reference produces / expects (tools/train.py:250-260 vs train_utils.py:87-116)."""
import os

import torch

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.checkpoint import find_last_checkpoint, load_saved_model, load_state_into

RNG = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]


def test_wrapper_and_raw_checkpoints_load_identically(tmp_path):
    args = synth.default_hypes(RNG)["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=9)
    run_a, run_b = tmp_path / "a", tmp_path / "b"
    run_a.mkdir(); run_b.mkdir()
    torch.save({"epoch": 4, "model_state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer_state_dict": {}},
               run_a / "net_epoch5.pth")                                   # what train.py writes (DDP-wrapped model)
    torch.save(sd, run_a / "net_epoch3.pth")
    torch.save(sd, run_b / "net_epoch12.pth")                              # a released raw state_dict
    assert find_last_checkpoint(str(run_a)) == 5 and find_last_checkpoint(str(run_b)) == 12
    assert find_last_checkpoint(str(tmp_path)) == 0
    ma, mb = Airv2xWhere2com(args), Airv2xWhere2com(args)
    ea, _ = load_saved_model(str(run_a), ma)
    eb, _ = load_saved_model(str(run_b), mb)
    assert (ea, eb) == (5, 12)
    for k, v in sd.items():
        assert torch.equal(ma.state_dict()[k], v) and torch.equal(mb.state_dict()[k], v), k
    e0, m0 = load_saved_model(str(tmp_path), Airv2xWhere2com(args))
    assert e0 == 0 and float(m0.state_dict()["cls_head.weight"].abs().sum()) == 0.0   # nothing to load: untouched


def test_mismatched_entries_keep_the_models_value():
    args = synth.default_hypes(RNG)["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=9)
    sd["cls_head.weight"] = torch.ones(3, 3)                               # wrong shape
    sd["not_a_parameter"] = torch.ones(1)
    del sd["reg_head.bias"]
    m = Airv2xWhere2com(args)
    rep = load_state_into(m, sd)
    assert rep["shape_mismatch"] == ["cls_head.weight"] and rep["dropped"] == ["not_a_parameter"] and rep["missing"] == ["reg_head.bias"]
    assert rep["loaded"] == len(m.state_dict()) - 2
    assert float(m.state_dict()["cls_head.weight"].abs().sum()) == 0.0
    assert torch.equal(m.state_dict()["obj_head.weight"], sd["obj_head.weight"])
