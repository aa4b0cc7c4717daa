"""GPU: the train-mode path of the camera branch (opencood_iface/train_camera.py, csrc/train_camera.hip, the lift adjoint in csrc/camera.hip;
SURVEY 8f #3 + #4).

* every new differentiable op against torch autograd of the same fp32 expression on the CPU: swish / sigmoid, add (+ ReLU), the depthwise
  convolution (k 3 / 5, stride 1 / 2, TF "same" padding) incl. its weight gradient, the squeeze-and-excite pieces, the bilinear
  upsample adjoint, the ground-truth-depth lift adjoint (against the oracle's voxel_pooling);
* one whole training step of Airv2xWhere2com WITH camera encoders -- camera + LiDAR (BASELINE configs[4]'s modality set) and camera only
  (the shipped camera YAML) -- against the REFERENCE's step (tests/golden/train_cam_small_*.npz: the reference's own model in .train()
  with stochastic depth off, its loss class, torch autograd; the float64 yardstick is the same reference model in double precision).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox
from tests.helpers import assert_close, load_fixture
from tests.test_gpu_train_when2com import _g, _loss, rel_close

pytestmark = pytest.mark.gpu


def test_unary_add_and_mean2():
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    g = _g(1)
    x, d = torch.randn(2, 5, 7, 32, generator=g) * 3, torch.randn(2, 5, 7, 32, generator=g)
    for fn, ref in ((TC.swish, lambda t: t * torch.sigmoid(t)), (TC.sigmoid, torch.sigmoid)):
        xr = x.clone().double().requires_grad_()
        ref(xr).backward(d.double())
        xd = x.cuda().requires_grad_()
        y = fn(xd)
        y.backward(d.cuda())
        rel_close(y.detach().cpu(), ref(x.double()), 2e-6, "unary")
        rel_close(xd.grad.cpu(), xr.grad, 2e-6, "unary grad")
    a, b = torch.randn(3, 4, 4, 64, generator=g), torch.randn(3, 4, 4, 64, generator=g)
    for relu in (False, True):
        ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
        yr = F.relu(ar + br) if relu else ar + br
        yr.backward(torch.ones_like(yr) * 0.5)
        ad, bd = a.cuda().requires_grad_(), b.cuda().requires_grad_()
        yd = TC.add_act(ad, bd, relu)
        yd.backward(torch.full_like(yd, 0.5))
        assert torch.equal(yd.detach().cpu(), yr.detach()) and torch.equal(ad.grad.cpu(), ar.grad) and torch.equal(bd.grad.cpu(), br.grad)
    ad, bd = a.cuda().requires_grad_(), b.cuda().requires_grad_()
    m = TC.Mean2Fn.apply(ad, bd)
    m.backward(torch.ones_like(m))
    rel_close(m.detach().cpu(), (a + b) / 2, 1e-6, "mean2")
    assert torch.equal(ad.grad.cpu(), torch.full_like(a, 0.5)) and torch.equal(bd.grad.cpu(), torch.full_like(a, 0.5))


@pytest.mark.parametrize("k,s,pad,h,w", [(3, 1, (1, 1), 9, 11), (5, 1, (2, 2), 8, 13), (3, 2, (0, 1), 12, 20), (5, 2, (1, 2), 13, 21), (3, 2, (1, 1), 11, 15)])
def test_depthwise_conv_forward_backward(k, s, pad, h, w):
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    n, c = 3, 64
    g = _g(10 * k + s)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, k, k, generator=g) / k
    xr, wr = x.clone().double().requires_grad_(), wt.clone().double().requires_grad_()
    yr = F.conv2d(F.pad(xr, (pad[0], pad[1], pad[0], pad[1])), wr, None, s, 0, 1, c)
    d = torch.randn(yr.shape, generator=g)
    yr.backward(d.double())
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    wd = wt.cuda().requires_grad_()
    yd = TC.DwConvFn.apply(xd, wd.reshape(c, k * k).t(), k, s, pad)
    yd.backward(d.permute(0, 2, 3, 1).contiguous().cuda())
    rel_close(yd.detach().cpu().permute(0, 3, 1, 2), yr.detach(), 2e-6, "dw forward")
    rel_close(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, 5e-6, "dw dx")
    rel_close(wd.grad.cpu(), wr.grad, 2e-5, "dw dw")


def test_squeeze_excite_pieces_and_batchnorm():
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    from airv2x_perception_amd.opencood_iface.train_when2com import linear_rows
    n, h, w, c, se = 11, 7, 9, 96, 4
    g = _g(5)
    x = torch.randn(n, c, h, w, generator=g)
    wr, br = torch.randn(se, c, generator=g) * 0.2, torch.randn(se, generator=g) * 0.1
    we, be = torch.randn(c, se, generator=g) * 0.2, torch.randn(c, generator=g) * 0.1
    gm, bt = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    d = torch.randn(n, c, h, w, generator=g)
    ref = [t.clone().double().requires_grad_() for t in (x, wr, br, we, be, gm, bt)]
    xb = F.batch_norm(ref[0], None, None, ref[5], ref[6], True, 0.0, 1e-3)
    gate = torch.sigmoid(F.linear((lambda t: t * torch.sigmoid(t))(F.linear(xb.mean((2, 3)), ref[1], ref[2])), ref[3], ref[4]))
    yr = xb * gate[:, :, None, None]
    yr.backward(d.double())
    dev = [t.cuda().requires_grad_() for t in (x.permute(0, 2, 3, 1).contiguous(), wr, br, we, be, gm, bt)]
    xbd = TC.BatchNormFn.apply(dev[0], dev[5], dev[6], 1e-3, None)
    gd = TC.sigmoid(linear_rows(TC.swish(linear_rows(TC.GapFn.apply(xbd), dev[1], dev[2], 0)), dev[3], dev[4], 0))
    yd = TC.ChannelScaleFn.apply(xbd, gd)
    yd.backward(d.permute(0, 2, 3, 1).contiguous().cuda())
    rel_close(yd.detach().cpu().permute(0, 3, 1, 2), yr.detach(), 1e-5, "se forward")
    rel_close(dev[0].grad.cpu().permute(0, 3, 1, 2), ref[0].grad, 5e-5, "se dx")
    for name, a, r in zip(("dwr", "dbr", "dwe", "dbe", "dgamma", "dbeta"), dev[1:], ref[1:]):
        rel_close(a.grad.cpu(), r.grad, 5e-5, name)


@pytest.mark.parametrize("scale,h,w,c", [(2, 7, 11, 64), (4, 4, 6, 256), (2, 1, 5, 32)])
def test_bilinear_upsample_adjoint(scale, h, w, c):
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    g = _g(scale * 100 + h)
    x = torch.randn(2, c, h, w, generator=g)
    xr = x.clone().double().requires_grad_()
    yr = F.interpolate(xr, scale_factor=scale, mode="bilinear", align_corners=True)
    d = torch.randn(yr.shape, generator=g)
    yr.backward(d.double())
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    yd = TC.ResizeFn.apply(xd, scale)
    yd.backward(d.permute(0, 2, 3, 1).contiguous().cuda())
    rel_close(yd.detach().cpu().permute(0, 3, 1, 2), yr.detach(), 2e-6, "resize forward")
    rel_close(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, 2e-6, "resize adjoint")


def test_lift_adjoint_is_the_gather_of_the_forward_scatter():
    """<LiftGt(feat), dout> == <feat, LiftGt^T(dout)> for random operands, with the training-mode clipping of the depth bins."""
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    from airv2x_perception_amd.opencood_iface.camera import CameraGeometry
    rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
    ca = synth.cam_args("vehicle", (104, 168), (rng[0], rng[3], rng[1], rng[4]))
    geo = CameraGeometry(ca, torch.device("cuda"))
    ci = synth.cam_inputs_for(7, 2, 2, (104, 168), "vehicle")
    flat = ci["imgs"].cuda().float().contiguous().view(4, 4, 104, 168)
    g = _g(9)
    feat = torch.randn(4, geo.fH, geo.fW, geo.C, generator=g).cuda().requires_grad_()
    pooled = TC.LiftGtFn.apply(feat, geo, flat, geo._cam_params(ci), 2, 2, True)
    dout = torch.randn(pooled.shape, generator=g).cuda()
    pooled.backward(dout)
    lhs = float((pooled.detach().double() * dout.double()).sum())
    rhs = float((feat.detach().double() * feat.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (lhs, rhs)
    assert float(feat.grad.abs().max()) > 0


@pytest.mark.parametrize("n,h,w,c", [(2, 52, 84, 64), (1, 7, 9, 32)])
def test_maxpool_backward_matches_torch(n, h, w, c):
    """nn.MaxPool2d(3, 2, 1) (the stem pool of CamEncode_Resnet101): forward bit-exact, the gradient lands on torch's positions (distinct values:
    no ties)."""
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    g = _g(h + w)
    x = torch.randn(n, c, h, w, generator=g)
    xr = x.clone().requires_grad_()
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    d = torch.randn(yr.shape, generator=g)
    yr.backward(d)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    yd = TC.MaxPoolFn.apply(xd, 3, 2, 1)
    yd.backward(d.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(yd.detach().cpu().permute(0, 3, 1, 2), yr.detach())
    assert torch.equal(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad)


def test_predicted_depth_lift_and_channel_softmax_adjoints():
    """The lift is bilinear in (features, depth distribution): <Lift(f, p), dout> = <f, df> = <p, dp>; the channel softmax's backward against
    torch autograd (CamEncode.get_depth_dist, lss_submodule.py:89-92) with padded logit channels."""
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    from airv2x_perception_amd.opencood_iface.camera import CameraGeometry
    rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
    ca = synth.cam_args("vehicle", (104, 168), (rng[0], rng[3], rng[1], rng[4]))
    geo = CameraGeometry(ca, torch.device("cuda"))
    ci = synth.cam_inputs_for(7, 2, 2, (104, 168), "vehicle")
    g = _g(11)
    feat = torch.randn(4, geo.fH, geo.fW, geo.C, generator=g).cuda().requires_grad_()
    dpad = (geo.nbins + 31) // 32 * 32
    logit = torch.randn(4, geo.fH, geo.fW, dpad, generator=g).cuda().requires_grad_()
    prob = TC.SoftmaxChFn.apply(logit, geo.nbins)
    prob.retain_grad()
    pooled = TC.LiftProbFn.apply(feat, prob, geo, geo._cam_params(ci), 2, 2)
    dout = torch.randn(pooled.shape, generator=g).cuda()
    pooled.backward(dout)
    lhs = float((pooled.detach().double() * dout.double()).sum())
    for name, a, b in (("features", feat, feat.grad), ("distribution", prob, prob.grad)):
        rhs = float((a.detach().double() * b.double()).sum())
        assert abs(lhs - rhs) <= 2e-5 * max(1.0, abs(lhs)), (name, lhs, rhs)
    lr = logit.detach().cpu().double().requires_grad_()
    pr = torch.softmax(lr[..., :geo.nbins], -1)
    pr.backward(prob.grad.cpu().double())
    rel_close(prob.detach().cpu(), pr.detach().float(), 2e-6, "softmax forward")
    rel_close(logit.grad.cpu(), lr.grad.float(), 2e-5, "softmax backward")
    assert float(logit.grad[..., geo.nbins:].abs().max()) == 0.0


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    final_dim = tuple(int(v) for v in fx["final_dim"])
    mods = tuple(str(m) for m in fx["modalities"])
    cams = {t: int(v) for t, v in zip(synth.AGENT_TYPES, fx["cams"])}
    hy = synth.multimodal_hypes(mods, rng, final_dim, bool(int(fx["use_depth_gt"])) if "use_depth_gt" in fx else True,
                                camera_encoder=str(fx["camera_encoder"]) if "camera_encoder" in fx else "EfficientNet")
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"])
            for i in range(len(types))]
    dd = synth.add_cameras(synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"]), types, seed=int(fx["seed"]) + 50,
                           final_dim=final_dim, cams_per_agent=cams)
    H, W = (int(v) for v in fx["head_hw"])
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    tgt = {k: torch.from_numpy(lc[k]).cuda() for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    return hy, args, sd, dd, tgt


@pytest.mark.parametrize("name", ["train_cam_small_n3", "train_cam_small_camonly_n2", "train_cam_small_camonly_n2b", "train_cam_small_softmax_n2",
                                  "train_cam_small_resnet101_n2"])
def test_camera_training_step_matches_the_reference(name, monkeypatch):
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface import train_camera as TC
    from airv2x_perception_amd.opencood_iface.train_where2com import forward_train
    monkeypatch.setattr(TC, "DROP_CONNECT", 0.0)          # the fixture's configuration edit: stochastic depth off
    fx = load_fixture(name)
    hy, args, sd, dd, tgt = _case(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    n, _, H, W = [int(v) for v in fx["mask_shape"]]
    ref_mask = torch.from_numpy(np.unpackbits(fx["mask"])[: n * H * W].reshape(n, H, W).astype(np.float32))
    K = [int(k) for k in fx["K"]]
    out = forward_train(model, dd, topk=K, mask=ref_mask)
    for k in ("psm", "rm", "obj"):
        assert out[k].requires_grad
        assert_close(out[k].detach().cpu(), fx[k], 1e-3, 1e-3 * float(np.abs(fx[k]).max()), k)
    total = _loss(args)(out, tgt)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - fx["losses"][0]) < 1e-3 * abs(fx["losses"][0])
    P = dict(model.named_parameters())
    keys = [str(k) for k in fx["grad_keys"]]
    noise = {k for k in keys if float(fx["g64max:" + k]) < 1e-12}
    missing = {k for k in keys if P[k].grad is None}
    assert missing <= noise, sorted(missing - noise)
    dev, refdev = {}, {}
    for k in keys:
        if k in noise:
            continue
        g = P[k].grad.reshape(-1)
        stride = int(fx["gsub"]) * max(1, g.numel() // 4096)
        gmax = float(fx["g64max:" + k])
        dev[k] = np.abs(g[::stride].cpu().numpy().astype(np.float64) - fx["g64:" + k].astype(np.float64)).max() / max(gmax, 1e-300)
        refdev[k] = float(fx["gdev:" + k])
    med_ref, med_dev = float(np.median(list(refdev.values()))), float(np.median(list(dev.values())))
    cam = [k for k in dev if ".camencode." in k or ".bevencode." in k]
    print(f"{name}: gradient deviation from float64, rel. to max -- device median {med_dev:.2e} worst {max(dev.values()):.2e}; "
          f"reference fp32 median {med_ref:.2e} worst {max(refdev.values()):.2e}; camera-branch tensors {len(cam)}: device median "
          f"{float(np.median([dev[k] for k in cam])):.2e}, reference {float(np.median([refdev[k] for k in cam])):.2e}")
    for grp in ("camencode", "bevencode", "backbone", "shrink", "cls_head", "reg_head"):
        ks = [k for k in dev if grp in k]
        if ks:
            print(f"   {grp:10s} {len(ks):4d} tensors: device median {float(np.median([dev[k] for k in ks])):.2e} worst {max(dev[k] for k in ks):.2e}"
                  f" | reference median {float(np.median([refdev[k] for k in ks])):.2e} worst {max(refdev[k] for k in ks):.2e}")
    assert len(cam) > 150
    # ReLU kinks (BevEncode's ResNet blocks, the BEV backbone) flip between ANY two fp32 evaluation orders, and one flip near the
    # loss moves every gradient below it by a few percent of its maximum (flip_count.py).  Three seeds, median deviation
    # device / reference: n3 2.7e-2 / 3.0e-2, camera-only n2 4.4e-2 / 2.8e-2 (uniformly from the backbone's block 1 down: a flip
    # there; its last conv weight is the worst tensor, 0.18 against the reference's 0.018), camera-only n2b 3.5e-2 / 3.9e-2 (there
    # the REFERENCE has the 0.16 outlier).  The heads and the shrink conv, with no kink between them and the loss, agree to
    # 3e-6 / 5e-4 / 1e-3 in all three.  Bound: twice the reference's own deviation on the median, 3x + 6 medians per tensor.
    bad = {k: (dev[k], refdev[k]) for k in dev if dev[k] > 3.0 * refdev[k] + 6.0 * med_ref + 1e-4}
    for k in bad:
        g = P[k].grad.reshape(-1)
        st = int(fx["gsub"]) * max(1, g.numel() // 4096)
        a, b = g[::st].cpu().numpy().astype(np.float64), fx["g64:" + k].astype(np.float64)
        print(f"   {k}: {a.size} samples, Frobenius-relative {np.linalg.norm(a - b) / np.linalg.norm(b):.3e}, entries off by > 1% of max: "
              f"{int((np.abs(a - b) > 0.01 * float(fx['g64max:' + k])).sum())}, rms g64/max {np.sqrt((b * b).mean()) / float(fx['g64max:' + k]):.3e}")
    assert not bad, bad
    assert med_dev <= 2.0 * med_ref + 1e-4, (med_dev, med_ref)
    smooth = [k for k in dev if "cls_head" in k or "reg_head" in k]
    assert smooth and all(dev[k] <= 2e-3 for k in smooth), {k: dev[k] for k in smooth}
    for k, b in model.named_buffers():
        ref = fx["b:" + k].astype(np.float64)
        assert np.abs(b.detach().cpu().numpy().astype(np.float64) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k


def test_camera_model_optimizer_steps_with_stochastic_depth():
    """The shipped stochastic depth (drop_connect 0.2), Adam steps, the loss goes down, .eval() runs the packed engine afterwards."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture("train_cam_small_camonly_n2")
    hy, args, sd, dd, tgt = _case(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    crit = _loss(args)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = crit(model(dd), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        o = model(dd)
    assert all(torch.isfinite(o[k]).all() for k in ("psm", "rm", "obj"))
