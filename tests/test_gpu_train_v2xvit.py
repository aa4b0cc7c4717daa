"""GPU: the train-mode path of Airv2xV2XVit (opencood_iface/train_v2xvit.py, train_fusion_ops.py, csrc/train_v2xvit.hip).

* the new differentiable ops (HGT attention on folded projections, the pyramid window attentions, split attention, warp, RTE add) against
  torch autograd of the oracle's fp32 expressions on the CPU;
* one whole training step against the REFERENCE's step (tests/golden/train_v2xvit_small_*.npz: the reference's own Airv2xV2XVit in .train()
  with the dropout probabilities set to 0, its loss class, torch autograd), float64 yardstick as in tests/test_gpu_train.py;
* the shipped dropout 0.3, optimiser steps, .eval() on the updated weights.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from airv2x_perception_amd import synth
from oracle import v2xvit_oracle as vit
from oracle import voxelize_oracle as vox
from tests.helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu
RNG = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]


def rel_close(got, ref, rtol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} (max|ref| {scale:.3e})"


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("types", [[0, 1, 1], [0, 0], [1, 0, 1, 0, 0]])
def test_hgt_attention_with_folded_relations_matches_autograd_of_the_oracle(types):
    """LN'd tokens -> per-type projections with relation_att / relation_msg folded in -> masked per-pixel attention over agents -> a_linears:
    the device graph (fold in tensor algebra on the weights, everything else HIP) against oracle.hgt_attention under torch autograd --
    gradients of the input AND of every parameter (relation matrices included)."""
    from airv2x_perception_amd.opencood_iface import train_fusion_ops as Fo
    from airv2x_perception_amd.opencood_iface.train_v2xvit import _groups, folded_projection
    n, H, W, C, heads, dh = len(types), 6, 8, 256, 8, 32
    g = _g(sum(types) * 10 + n)
    x = torch.randn(n, H, W, C, generator=g)
    mask = (torch.rand(n, H, W, generator=g) > 0.3).float()
    mask[0] = 1.0
    dy = torch.randn(n, H, W, C, generator=g)
    h = "h"
    sd = {h + ".relation_att": torch.randn(4, heads, dh, dh, generator=g) * 0.2, h + ".relation_msg": torch.randn(4, heads, dh, dh, generator=g) * 0.2}
    for name in ("q", "k", "v", "a"):
        for t in range(2):
            sd[f"{h}.{name}_linears.{t}.weight"] = torch.randn(C, C, generator=g) * 0.06
            sd[f"{h}.{name}_linears.{t}.bias"] = torch.randn(C, generator=g) * 0.1
    ref = {k: v.clone().requires_grad_() for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    tt = torch.tensor([types])
    yo = vit.hgt_attention(xr.unsqueeze(0), mask.permute(1, 2, 0).unsqueeze(0).unsqueeze(3), tt, ref, h, heads, dh)[0]
    yo.backward(dy)
    P = {k: v.clone().cuda().requires_grad_() for k, v in sd.items()}
    xd = x.cuda().requires_grad_()
    groups = _groups(types)
    fold = {t: folded_projection(P, h, t, heads, dh) for t in set(types)}
    proj = torch.cat([Fo.linear(xd[a:b], fold[t][0], fold[t][1]) for (a, b, t) in groups], 0)
    att = Fo.hgt_attention(proj, mask.cuda(), types, heads, dh)
    yd = torch.cat([Fo.linear(att[a:b], P[f"{h}.a_linears.{t}.weight"], P[f"{h}.a_linears.{t}.bias"]) for (a, b, t) in groups], 0)
    yd.backward(dy.cuda())
    rel_close(yd.detach().cpu(), yo.detach(), 5e-5, "hgt forward")
    rel_close(xd.grad.cpu(), xr.grad, 1e-4, "hgt dx")
    for k in sd:
        if ref[k].grad is None:
            assert P[k].grad is None or float(P[k].grad.abs().max()) == 0.0, k
            continue
        if float(ref[k].grad.abs().max()) < 1e-7 * float(ref[h + ".relation_msg"].grad.abs().max()):
            continue        # exactly-zero directions (a key bias shared by every key of a pixel cancels in the softmax): rounding noise only
        rel_close(P[k].grad.cpu(), ref[k].grad, 2e-4, k)


def test_pyramid_window_attention_backward_matches_autograd_of_the_oracle():
    from airv2x_perception_amd.opencood_iface import train_fusion_ops as Fo
    n, H, W, C = 2, 8, 12, 256
    cfg = [(16, 16, 2), (8, 32, 4), (4, 64, 4)]
    g = _g(7)
    x = torch.randn(1, n, H, W, C, generator=g)
    sd, douts = {}, []
    for i, (h, dh, ws) in enumerate(cfg):
        sd[f"w.{i}.to_qkv.weight"] = torch.randn(3 * h * dh, C, generator=g) * 0.08
        sd[f"w.{i}.pos_embedding"] = torch.randn(2 * ws - 1, 2 * ws - 1, generator=g)
        sd[f"w.{i}.to_out.0.weight"] = torch.randn(C, h * dh, generator=g) * 0.08
        sd[f"w.{i}.to_out.0.bias"] = torch.randn(C, generator=g) * 0.1
        douts.append(torch.randn(1, n, H, W, C, generator=g))
    ref = {k: v.clone().requires_grad_() for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    outs_r = [vit.window_attention(xr, ref, f"w.{i}", h, dh, ws) for i, (h, dh, ws) in enumerate(cfg)]
    sum((o * d).sum() for o, d in zip(outs_r, douts)).backward()
    P = {k: v.clone().cuda().requires_grad_() for k, v in sd.items()}
    xd = x[0].cuda().requires_grad_()
    qkv3 = Fo.linear(xd, torch.cat([P[f"w.{i}.to_qkv.weight"] for i in range(3)], 0))
    wat = Fo.pyramid_window_attention(qkv3, [P[f"w.{i}.pos_embedding"] for i in range(3)], cfg)
    outs_d = [Fo.linear(wat[i], P[f"w.{i}.to_out.0.weight"], P[f"w.{i}.to_out.0.bias"]) for i in range(3)]
    sum((o * d[0].cuda()).sum() for o, d in zip(outs_d, douts)).backward()
    for i in range(3):
        rel_close(outs_d[i].detach().cpu(), outs_r[i].detach()[0], 5e-5, f"window branch {i}")
    rel_close(xd.grad.cpu(), xr.grad[0], 1e-4, "window dx")
    for k in sd:
        rel_close(P[k].grad.cpu(), ref[k].grad, 2e-4, k)


def test_split_attention_warp_and_rte_backward():
    from airv2x_perception_amd.opencood_iface import train_fusion_ops as Fo
    from airv2x_perception_amd.opencood_iface import warp as warp_host
    n, H, W, C = 3, 8, 12, 256
    g = _g(11)
    br = [torch.randn(1, n, H, W, C, generator=g) for _ in range(3)]
    res = torch.randn(n, H, W, C, generator=g)
    sd = {"s.fc1.weight": torch.randn(C, C, generator=g) * 0.1, "s.bn1.weight": torch.rand(C, generator=g) + 0.5, "s.bn1.bias": torch.randn(C, generator=g) * 0.1,
          "s.fc2.weight": torch.randn(3 * C, C, generator=g) * 0.3}
    dy = torch.randn(n, H, W, C, generator=g)
    ref = {k: v.clone().requires_grad_() for k, v in sd.items()}
    brr = [b.clone().requires_grad_() for b in br]
    rr = res.clone().requires_grad_()
    (vit.split_attn(brr, ref, "s")[0] + rr).backward(dy)
    P = {k: v.clone().cuda().requires_grad_() for k, v in sd.items()}
    brd = [b[0].cuda().requires_grad_() for b in br]
    rd = res.cuda().requires_grad_()
    yd = Fo.split_attn(brd[0], brd[1], brd[2], rd, P["s.fc1.weight"], P["s.bn1.weight"], P["s.bn1.bias"], P["s.fc2.weight"])
    yd.backward(dy.cuda())
    rel_close(yd.detach().cpu(), (vit.split_attn(br, sd, "s")[0] + res), 2e-5, "split-attn forward")
    for i in range(3):
        rel_close(brd[i].grad.cpu(), brr[i].grad[0], 1e-4, f"split-attn d branch {i}")
    assert torch.equal(rd.grad.cpu(), dy)
    for k in sd:
        rel_close(P[k].grad.cpu(), ref[k].grad, 2e-4, k)
    # warp_affine: gradient = the adjoint of the bilinear sampling (scatter), against autograd of grid_sample
    x = torch.randn(2, C, H, W, generator=g)
    M = torch.tensor([[[0.98, -0.17, 1.3], [0.17, 0.98, -0.8]], [[1.0, 0.05, -2.2], [-0.05, 1.0, 0.6]]])
    xr = x.clone().requires_grad_()
    yr = vit.warp_affine(xr, M, (H, W))
    dyw = torch.randn(yr.shape, generator=g)
    yr.backward(dyw)
    theta = torch.from_numpy(warp_host.affine_theta(M.numpy(), (H, W), (H, W))).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    ydw = Fo.warp_affine(xd, theta)
    ydw.backward(dyw.permute(0, 2, 3, 1).contiguous().cuda())
    rel_close(ydw.detach().cpu().permute(0, 3, 1, 2), yr.detach(), 2e-5, "warp forward")
    rel_close(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, 2e-5, "warp adjoint")
    xd2 = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    Fo.warp_affine(xd2, theta).backward(dyw.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(xd2.grad, xd.grad)          # fixed-point scatter: bit-reproducible
    # RTE add: dv = per-agent sums over the map
    xa, va = torch.randn(3, 4, 6, C, generator=g), torch.randn(3, C, generator=g)
    xd, vd = xa.cuda().requires_grad_(), va.cuda().requires_grad_()
    ya = Fo.add_agent_vector(xd, vd)
    da = torch.randn(3, 4, 6, C, generator=g)
    ya.backward(da.cuda())
    assert_close(ya.detach().cpu(), xa + va[:, None, None, :], 1e-6, 1e-6, "rte add")
    rel_close(vd.grad.cpu(), da.sum((1, 2)), 2e-5, "rte dv")
    assert torch.equal(xd.grad.cpu(), da)


def _case(fx, dropout=0.0):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_v2xvit(rng, tuple(int(v) for v in fx["max_cav"]))
    e = hy["model"]["args"]["transformer"]["encoder"]
    e["cav_att_config"]["dropout"] = e["pwindow_att_config"]["dropout"] = e["feed_forward"]["dropout"] = dropout
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.v2xvit_param_spec(args), seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"])
            for i in range(len(types))]
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    dd["spatial_correction_matrix"] = torch.from_numpy(fx["spatial_correction_matrix"])
    dd["prior_encoding"] = torch.from_numpy(fx["prior_encoding"])
    H, W = (int(v) for v in fx["head_hw"]) if "head_hw" in fx else fx["psm"].shape[-2:]    # full-grid fixtures store strided heads
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    tgt = {k: torch.from_numpy(lc[k]).cuda() for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    return hy, args, sd, dd, tgt


def _model(args, sd):
    from airv2x_perception_amd.opencood_iface import Airv2xV2XVit
    m = Airv2xV2XVit(args)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _loss(args):
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    return PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})


@pytest.mark.parametrize("name", ["train_v2xvit_small_n3", "train_v2xvit_small_n2", "train_v2xvit_full_n4"])
def test_v2xvit_training_step_matches_the_reference(name):
    fx = load_fixture(name)
    hy, args, sd, dd, tgt = _case(fx)
    model = _model(args, sd)
    out = model(dd)
    for k in ("psm", "rm", "obj"):
        assert out[k].requires_grad
        hs = int(fx["head_stride"]) if "head_stride" in fx else 1
        assert_close(out[k].detach().cpu()[..., ::hs, ::hs], fx[k], 3e-4, 3e-4 * float(np.abs(fx[k]).max()), k)
    total = _loss(args)(out, tgt)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - fx["losses"][0]) < 3e-4 * abs(fx["losses"][0])
    P = dict(model.named_parameters())
    keys = [str(k) for k in fx["grad_keys"]]
    have = sorted(k for k, p in P.items() if p.grad is not None)
    assert set(keys) <= set(have), sorted(set(keys) - set(have))
    gscale = float(np.median([float(fx["g64max:" + k]) for k in keys]))
    dev, refdev = {}, {}
    for k in keys:
        g = P[k].grad.reshape(-1)
        stride = max(1, g.numel() // 4096)
        # relative to the tensor's largest exact entry -- floored: a key bias shared by all keys of a pixel has an EXACTLY zero gradient
        # (it cancels in the softmax; float64 gives 1e-20), where "relative" error of any fp32 evaluation is meaningless
        gmax = max(float(fx["g64max:" + k]), 1e-6 * gscale)
        dev[k] = np.abs(g[::stride].cpu().numpy().astype(np.float64) - fx["g64:" + k].astype(np.float64)).max() / gmax
        refdev[k] = np.abs(fx["g:" + k].astype(np.float64) - fx["g64:" + k].astype(np.float64)).max() / gmax
    med_ref, med_dev = float(np.median(list(refdev.values()))), float(np.median(list(dev.values())))
    print(f"{name}: gradient deviation from float64, rel. to max -- device median {med_dev:.2e} worst {max(dev.values()):.2e}; "
          f"reference fp32 median {med_ref:.2e} worst {max(refdev.values()):.2e}")
    bad = {k: (dev[k], refdev[k]) for k in keys if dev[k] > 3.0 * refdev[k] + 2.0 * med_ref + 2e-4}
    assert not bad, bad
    assert med_dev <= 1.5 * med_ref + 2e-4, (med_dev, med_ref)
    assert max(dev.values()) <= 2.5 * max(refdev.values()) + 2e-4, (max(dev.values()), max(refdev.values()))
    for k, b in model.named_buffers():
        ref = fx["b:" + k].astype(np.float64)
        assert np.abs(b.detach().cpu().numpy().astype(np.float64) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k


def test_v2xvit_optimizer_steps_dropout_and_eval():
    fx = load_fixture("train_v2xvit_small_n3")
    hy, args, sd, dd, tgt = _case(fx, dropout=0.3)           # the shipped YAML's dropout
    model = _model(args, sd)
    model.sync_comm_rate = False
    crit = _loss(args)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=2e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(model(dd), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        o1 = model(dd)
        sd_now = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o2 = vit.v2xvit_forward(dd, sd_now, args)
    for k in ("psm", "rm", "obj"):
        assert_close(o1[k].cpu(), o2[k], 1e-3, 1e-3 * float(o2[k].abs().max()), k)
