"""Detection loss (reference: loss/point_pillar_loss_multiclass.py:79-179).  tests/golden/loss_small.npz holds the losses
and the autograd gradients of the REFERENCE's own class on seeded head maps / labels (tools/gen_golden.py loss; one case
with a sample without positives, one NaN regression target per case).  CPU: the oracle restatement is bit-equal to them.
GPU: ``opencood_iface.loss.PointPillarLossMultiClass`` (av2x_pp_loss: fused forward + gradient) within
1e-5 relative on the scalars (fp64 partial sums vs torch's fp32 tree) and 1e-6 + 2e-5 * |ref| on the gradients
(expf / log1pf vs ATen's CPU vector math)."""
import os

import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loss_small.npz")
CASES = (("a", dict(seed=5)), ("b", dict(seed=6, B=3, H=7, W=9, empty_sample=1)), ("c", dict(seed=7, B=1, H=25, W=44, pos_frac=0.004)))
ARGS = {"cls_weight": 1.0, "reg": 2.0, "num_class": 7}


@pytest.mark.parametrize("tag,kw", CASES)
def test_oracle_equals_reference(tag, kw):
    from oracle import loss_oracle as lo
    g = np.load(GOLD)
    t = {k: torch.from_numpy(v) for k, v in synth.loss_case(**kw).items()}
    heads = {k: t[k].clone().requires_grad_(True) for k in ("psm", "rm", "obj")}
    total, reg, conf, _ = lo.pp_loss(heads["psm"], heads["rm"], heads["obj"], t["targets"], t["pos_equal_one"], t["class_ids"], 7, 1.0, 2.0)
    total.backward()
    np.testing.assert_array_equal(np.asarray([float(total.detach()), float(reg.detach()), float(conf.detach())]), g[f"{tag}_losses"])
    for k in ("psm", "rm", "obj"):
        np.testing.assert_array_equal(heads[k].grad.numpy(), g[f"{tag}_d{k}"])


def test_mirror_contract_on_cpu():
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    crit = PointPillarLossMultiClass(ARGS)
    assert (crit.cls_weight, crit.reg_coe, crit.cls_num, crit.alpha, crit.gamma) == (1.0, 2.0, 7, 0.25, 2.0)
    assert list(crit.state_dict().keys()) == []
    t = {k: torch.from_numpy(v) for k, v in synth.loss_case(seed=5).items()}
    with pytest.raises(RuntimeError, match="no CPU path"):
        crit(t, t)
    crit.loss_dict.update({"total_loss": 1.25, "reg_loss": 0.5, "conf_loss": 0.75})
    assert crit.logging(3, 9, 100) == "[epoch 3][10/100], || Loss: 1.25 ||total: 1.25 | reg: 0.50 | conf: 0.75 | "


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", CASES)
def test_gpu_loss_and_gradients_match_reference(tag, kw):
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    g = np.load(GOLD)
    t = {k: torch.from_numpy(v).cuda() for k, v in synth.loss_case(**kw).items()}
    heads = {k: t[k].clone().requires_grad_(True) for k in ("psm", "rm", "obj")}
    crit = PointPillarLossMultiClass(ARGS)
    total = crit(heads, {k: t[k] for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")})
    ref = g[f"{tag}_losses"]
    got = np.asarray([float(total.detach()), crit.loss_dict["reg_loss"], crit.loss_dict["conf_loss"]])
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref)), (got, ref)
    (2.0 * total).backward()          # an upstream factor must scale the gradients
    for k in ("psm", "rm", "obj"):
        r = 2.0 * g[f"{tag}_d{k}"]
        d = np.abs(heads[k].grad.cpu().numpy() - r)
        assert np.all(d <= 1e-6 + 2e-5 * np.abs(r)), (k, float(d.max()))
    # forward only (validation loss): no gradient buffers, same value, prefix handling
    with torch.no_grad():
        v = crit({k + "_single": t[k] for k in ("psm", "rm", "obj")},
                 {k: t[k] for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}, prefix="_single")
    assert float(v) == float(total.detach()) and "total_loss_single" in crit.loss_dict


@pytest.mark.gpu
def test_gpu_loss_dict_is_read_back_lazily_and_an_out_of_range_class_id_raises_at_the_read():
    """forward() queues the three scalars (and the class-id range of point_pillar_loss_multiclass.py:118-125's one_hot) to a pinned buffer
    without waiting for the device; the floats -- and the IndexError the reference's scatter_ would raise -- appear at the first read."""
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    tag, kw = CASES[0]
    t = {k: torch.from_numpy(v).cuda() for k, v in synth.loss_case(**kw).items()}
    heads = {k: t[k].clone().requires_grad_(True) for k in ("psm", "rm", "obj")}
    tgt = {k: t[k] for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    crit = PointPillarLossMultiClass(ARGS)
    total = crit(heads, tgt)
    assert crit.loss_dict._pending and not dict.__len__(crit.loss_dict)          # nothing read back yet
    total.backward()
    assert abs(crit.loss_dict["total_loss"] - float(total.detach())) <= 1e-6 * abs(float(total.detach()))
    assert not crit.loss_dict._pending
    assert "Loss" in crit.logging(0, 0, 1)
    bad = dict(tgt)
    bad["class_ids"] = tgt["class_ids"].clone()
    bad["class_ids"].view(-1)[3] = ARGS["num_class"]
    crit(heads, bad)
    with pytest.raises(IndexError):
        crit.loss_dict["total_loss"]
    crit.validate_class_ids = "sync"            # strict mode: raises before forward() returns, where the reference's one_hot scatter_ does
    with pytest.raises(IndexError, match="out of range"):
        crit(heads, bad)
    assert float(crit(heads, tgt).detach()) == float(total.detach())
    crit.validate_class_ids = False
    crit(heads, bad)
    assert crit.loss_dict["total_loss"] == crit.loss_dict["total_loss"]


@pytest.mark.gpu
def test_gpu_loss_full_head_maps():
    """Two frames of full 100 x 352 head maps: the three scalars and strided samples / abs-sums of the gradients."""
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    g = np.load(GOLD)
    t = {k: torch.from_numpy(v).cuda() for k, v in synth.loss_case(seed=8, B=2, H=100, W=352, pos_frac=0.002).items()}
    heads = {k: t[k].clone().requires_grad_(True) for k in ("psm", "rm", "obj")}
    crit = PointPillarLossMultiClass(ARGS)
    total = crit(heads, {k: t[k] for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")})
    ref = g["full_losses"]
    got = np.asarray([float(total.detach()), crit.loss_dict["reg_loss"], crit.loss_dict["conf_loss"]])
    assert np.all(np.abs(got - ref) <= 2e-5 * np.abs(ref)), (got, ref)   # torch sums 2 M fp32 terms pairwise; the kernel in fp64
    total.backward()
    for k in ("psm", "rm", "obj"):
        gr = heads[k].grad
        r = g[f"full_d{k}"]
        d = np.abs(gr[:, :, ::7, ::11].cpu().numpy() - r)
        assert np.all(d <= 1e-7 + 2e-5 * np.abs(r)), (k, float(d.max()))
        s_ref = float(g[f"full_d{k}_abs_sum"])
        assert abs(float(gr.double().abs().sum()) - s_ref) <= 1e-5 * s_ref, k
