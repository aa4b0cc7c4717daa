"""GPU: the train-mode path of Airv2xWhen2com (opencood_iface/train_when2com.py, csrc/train_when2com.hip; SURVEY 8f #2 + #4).

* every new differentiable op against torch autograd of the oracle's fp32 expression on the CPU (warp_affine_simple, the row GEMM of
  km_generator incl. its ReLU / bias / more than 8 rows, the softmax-over-keys attention);
* one whole training step -- forward in train mode, PointPillarLossMultiClass, backward, BatchNorm running statistics (trunk momentum 0.01,
  policy_net4's nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1, conv bias folded into the running mean) -- against the REFERENCE's step
  (tests/golden/train_when2com_small_*.npz: the reference's own Airv2xWhen2com in .train(), its loss class, torch autograd), with the
  float64 yardstick of the Where2Comm / CoBEVT step tests;
* optimiser steps, .eval() on the updated weights.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox
from oracle import when2com_oracle as w2
from tests.helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu


def rel_close(got, ref, rtol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} (max|ref| {scale:.3e})"


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("n,H,W,C", [(3, 12, 20, 256), (1, 7, 9, 64), (2, 16, 16, 128)])
def test_warp_affine_simple_forward_backward(n, H, W, C):
    from airv2x_perception_amd.opencood_iface import train_when2com as Tw
    g = _g(n * 100 + H)
    x = torch.randn(n, C, H, W, generator=g)
    th = torch.eye(2, 3).repeat(n, 1, 1)
    for j in range(n):
        a = 0.15 * j - 0.1
        th[j] = torch.tensor([[np.cos(a), -np.sin(a) * H / W, 0.2 * j - 0.1], [np.sin(a) * W / H, np.cos(a), 0.05 - 0.15 * j]])
    dy = torch.randn(n, C, H, W, generator=g)
    xr = x.clone().requires_grad_()
    yr = w2.warp_affine_simple(xr, th, (H, W))
    yr.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    yd = Tw.warp_affine_simple(xd, th.cuda().contiguous())
    yd.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    assert_close(yd.detach().cpu().permute(0, 3, 1, 2), yr.detach(), 2e-5, 2e-5, "warp forward")
    rel_close(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, 2e-5, "warp dx")


def test_warp_adjoint_gather_and_scatter_paths_agree_with_autograd():
    """The adjoint of the warp is a gather over the output pixels around M^-1 (s - C) for every well-conditioned theta and, per image, the
    fixed-point scatter for the rest (zoom-in by 10: hundreds of output pixels per source pixel; a singular theta; all zeros): both against
    torch autograd of grid_sample, in ONE launch with mixed images; a zoom-out, a shear and a flip take the gather."""
    from airv2x_perception_amd.opencood_iface import train_when2com as Tw
    from airv2x_perception_amd.opencood_iface import train_fusion_ops as Fo
    g = _g(77)
    H, W, C = 20, 28, 128
    ths = [[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], [[0.1, 0.0, 0.3], [0.0, 0.1, -0.2]], [[3.0, 0.4, 0.1], [-0.3, 2.5, 0.2]],
           [[1.0, 1.0, 0.0], [1.0, 1.0, 0.0]], [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]], [[-1.0, 0.3, 0.05], [0.1, -0.9, -0.1]],
           [[0.7, -0.7 * H / W, 0.4], [0.7 * W / H, 0.7, -0.6]], [[1.0, 0.0, 2.5], [0.0, 1.0, 0.0]]]
    th = torch.tensor(ths, dtype=torch.float32)
    n = th.shape[0]
    x = torch.randn(n, C, H, W, generator=g)
    dy = torch.randn(n, C, H, W, generator=g)
    for simple in (True, False):
        xr = x.clone().requires_grad_()
        grid = F.affine_grid(th, [n, C, H, W], align_corners=not simple)
        F.grid_sample(xr, grid, align_corners=not simple).backward(dy)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
        op = Tw.warp_affine_simple if simple else Fo.warp_affine
        op(xd, th.cuda().contiguous()).backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
        got, ref = xd.grad.cpu().permute(0, 3, 1, 2), xr.grad
        for k in range(n):
            rel_close(got[k], ref[k], 3e-5, f"warp adjoint, simple={simple}, theta {k}")
        xd2 = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
        op(xd2, th.cuda().contiguous()).backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
        assert torch.equal(xd2.grad, xd.grad)          # fixed candidate order / fixed-point sums: bit-reproducible


@pytest.mark.parametrize("m,n,k,act", [(3, 256, 4096, 1), (1, 128, 256, 1), (7, 32, 128, 0), (11, 64, 1024, 1), (2, 256, 50688, 1)])
def test_linear_rows_forward_backward(m, n, k, act):
    from airv2x_perception_amd.opencood_iface import train_when2com as Tw
    g = _g(m * 1000 + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    b = torch.randn(n, generator=g) * 0.1
    dy = torch.randn(m, n, generator=g)
    ref = [t.clone().double().requires_grad_() for t in (x, w, b)]
    yr = F.linear(ref[0], ref[1], ref[2])
    yr = F.relu(yr) if act else yr
    yr.backward(dy.double())
    dev = [t.cuda().requires_grad_() for t in (x, w, b)]
    if m > 8:   # the bias gradient is formed for <= 8 rows (one row per agent): no bias there
        dev[2] = None
        yr2 = F.linear(ref[0].detach(), ref[1].detach())
        yd = Tw.linear_rows(dev[0], dev[1], None, act)
        rel_close(yd.detach().cpu(), (F.relu(yr2) if act else yr2), 2e-5, "forward (no bias)")
        return
    yd = Tw.linear_rows(dev[0], dev[1], dev[2], act)
    yd.backward(dy.cuda())
    rel_close(yd.detach().cpu(), yr.detach(), 2e-5, "forward")
    for name, a, r in zip(("dx", "dw", "db"), dev, ref):
        rel_close(a.grad.cpu(), r.grad, 3e-5, name)
    # run-to-run identical
    dev2 = [t.cuda().requires_grad_() for t in (x, w, b)]
    Tw.linear_rows(dev2[0], dev2[1], dev2[2], act).backward(dy.cuda())
    assert all(torch.equal(a.grad, c.grad) for a, c in zip(dev, dev2))


@pytest.mark.parametrize("n,ks", [(3, 256), (1, 32), (5, 64)])
def test_attention_over_the_keys_forward_backward(n, ks):
    from airv2x_perception_amd.opencood_iface import train_when2com as Tw
    g = _g(n * 10 + ks)
    H, W, C = 6, 10, 64
    keys, q = torch.randn(n, ks, generator=g) * 0.3, torch.randn(1, ks, generator=g) * 0.3
    maps = torch.randn(n, H, W, C, generator=g)
    dout = torch.randn(1, H, W, C, generator=g)
    ref = [t.clone().double().requires_grad_() for t in (keys, q, maps)]
    p = torch.softmax(ref[0] @ ref[1].t(), dim=0)
    yr = (p.view(n, 1, 1, 1) * ref[2]).sum(0, keepdim=True)
    yr.backward(dout.double())
    dev = [t.cuda().requires_grad_() for t in (keys, q, maps)]
    yd = Tw.when2com_attention(*dev)
    yd.backward(dout.cuda())
    rel_close(yd.detach().cpu(), yr.detach(), 1e-5, "fused")
    for name, a, r in zip(("dkeys", "dq", "dmaps"), dev, ref):
        rel_close(a.grad.cpu(), r.grad, 3e-5, name)


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_when2com(rng)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.when2com_param_spec(args), seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"])
            for i in range(len(types))]
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    dd["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(len(types), args["max_cav_num"])
    H, W = (int(v) for v in fx["head_hw"])
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    tgt = {k: torch.from_numpy(lc[k]).cuda() for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    return hy, args, sd, dd, tgt


def _model(args, sd):
    from airv2x_perception_amd.opencood_iface import Airv2xWhen2com
    m = Airv2xWhen2com(args)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _loss(args):
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    return PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})


@pytest.mark.parametrize("name", ["train_when2com_small_n3", "train_when2com_small_n2", "train_when2com_full_n3"])
def test_when2com_training_step_matches_the_reference(name):
    fx = load_fixture(name)
    hy, args, sd, dd, tgt = _case(fx)
    model = _model(args, sd)
    out = model(dd)
    # communication_rates counts the non-zeros of the shrink header's ReLU output: discontinuous at 0, a last-bit difference moves it by one
    assert abs(float(out["comm_rate"]) - float(fx["comm_rate"])) <= max(2.0, 1e-5 * float(fx["comm_rate"])) and out["mask"] == 0
    for k in ("psm", "rm", "obj"):
        assert out[k].requires_grad
        hs = int(fx["head_stride"])
        assert_close(out[k].detach().cpu()[..., ::hs, ::hs], fx[k], 3e-4, 3e-4 * float(np.abs(fx[k]).max()), k)
    total = _loss(args)(out, tgt)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - fx["losses"][0]) < 3e-4 * abs(fx["losses"][0])
    P = dict(model.named_parameters())
    keys = [str(k) for k in fx["grad_keys"]]
    have = sorted(k for k, p in P.items() if p.grad is not None)
    # (the policy-net conv biases have no path in this build's graph -- gradient None = the exact value, zero; see `noise` below)
    assert set(keys) - set(have) <= {k for k in keys if float(fx["g64max:" + k]) < 1e-12}, sorted(set(keys) - set(have))
    for k in set(have) - set(keys):      # parameters the reference reports an exactly-zero gradient for
        assert float(P[k].grad.abs().max()) == 0.0, k
    dev, refdev = {}, {}
    for k in keys:
        if P[k].grad is None:
            continue
        g = P[k].grad.reshape(-1)
        stride = max(1, g.numel() // 4096)
        gmax = float(fx["g64max:" + k])
        dev[k] = np.abs(g[::stride].cpu().numpy().astype(np.float64) - fx["g64:" + k].astype(np.float64)).max() / max(gmax, 1e-300)
        refdev[k] = float(fx["gdev:" + k])
    med_ref, med_dev = float(np.median(list(refdev.values()))), float(np.median(list(dev.values())))
    print(f"{name}: gradient deviation from float64, rel. to max -- device median {med_dev:.2e} worst {max(dev.values()):.2e}; "
          f"reference fp32 median {med_ref:.2e} worst {max(refdev.values()):.2e}")
    # parameters whose TRUE gradient is zero -- the conv biases of policy_net4 (in front of a batch-statistics BatchNorm) and the last
    # key_net bias (a constant added to every key shifts all logits alike: the softmax does not see it): the float64 step gives < 1e-12,
    # the reference's fp32 autograd its own rounding noise.  The device's must be noise of that order too; everything else is compared.
    noise = {k for k in keys if float(fx["g64max:" + k]) < 1e-12}
    assert any(".cbr_unit.0.bias" in k for k in noise) and "fusion_net.key_net.fc.4.bias" in noise
    cmp_keys = [k for k in keys if k not in noise]
    for k in noise:
        if P[k].grad is None:
            continue
        assert float(P[k].grad.abs().max()) <= max(1e-6, 100.0 * float(fx["gsum:" + k][2])), (k, float(P[k].grad.abs().max()), fx["gsum:" + k][2])
    assert all(P[k].grad is not None for k in cmp_keys)
    refdev = {k: refdev[k] for k in cmp_keys}
    dev = {k: dev[k] for k in cmp_keys}
    med_ref, med_dev = float(np.median(list(refdev.values()))), float(np.median(list(dev.values())))
    bad = {k: (dev[k], refdev[k]) for k in cmp_keys if dev[k] > 3.0 * refdev[k] + 2.0 * med_ref + 1e-4}
    assert not bad, bad
    assert med_dev <= 1.5 * med_ref + 1e-4, (med_dev, med_ref)
    for k, b in model.named_buffers():
        ref = fx["b:" + k].astype(np.float64)
        assert np.abs(b.detach().cpu().numpy().astype(np.float64) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k


def test_when2com_optimizer_steps_and_eval():
    """Adam steps, the loss goes down, and .eval() runs the packed engine on the updated weights."""
    fx = load_fixture("train_when2com_small_n2")
    hy, args, sd, dd, tgt = _case(fx)
    model = _model(args, sd)
    crit = _loss(args)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(model(dd), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.95 * losses[0], losses
    model.eval()
    with torch.no_grad():
        o1 = model(dd)
        sd_now = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o2 = w2.when2com_forward(dd, sd_now, args)
    for k in ("psm", "rm", "obj"):
        assert_close(o1[k].cpu(), o2[k], 1e-3, 1e-3 * float(o2[k].abs().max()), k)
