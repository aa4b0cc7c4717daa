"""GPU: the train-mode path of Airv2xV2VNet (opencood_iface/train_v2vnet.py, csrc/train_when2com.hip; SURVEY 8f #2 + #4).

* the new differentiable ops against torch autograd on the CPU (the one-step ConvGRU gate, the max over the neighbours' messages);
* one whole training step against the REFERENCE's step (tests/golden/train_v2vnet_small_*.npz: the reference's own Airv2xV2VNet in
  .train(), agg_operator avg and max, its loss class, torch autograd), float64 yardstick as for the other models: the reset-gate rows
  and hidden-state input columns of the ConvGRU weights must get the exact-zero gradients the reference's autograd gives them;
* optimiser steps, .eval() on the updated weights.
"""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import v2vnet_oracle as v2v
from oracle import voxelize_oracle as vox
from tests.helpers import assert_close, load_fixture
from tests.test_gpu_train_when2com import _g, _loss, rel_close

pytestmark = pytest.mark.gpu


def test_gru_gate_and_agent_max_forward_backward():
    from airv2x_perception_amd.opencood_iface import train_v2vnet as Tv
    g = _g(3)
    beta, cnm, dout = (torch.randn(1, 6, 10, 256, generator=g) * 2 for _ in range(3))
    ref = [t.clone().double().requires_grad_() for t in (beta, cnm)]
    yr = torch.sigmoid(ref[0]) * torch.tanh(ref[1])
    yr.backward(dout.double())
    dev = [t.cuda().requires_grad_() for t in (beta, cnm)]
    yd = Tv.GruGateFn.apply(*dev)
    yd.backward(dout.cuda())
    rel_close(yd.detach().cpu(), yr.detach(), 1e-6, "gate")
    for name, a, r in zip(("dbeta", "dcnm"), dev, ref):
        rel_close(a.grad.cpu(), r.grad, 2e-6, name)
    x = torch.randn(4, 5, 7, 64, generator=g)
    x[2] = x[0]          # ties: the first maximum keeps the gradient
    d = torch.randn(1, 5, 7, 64, generator=g)
    xr = x.clone().requires_grad_()
    mr = xr.max(0, keepdim=True)[0]
    xd = x.cuda().requires_grad_()
    md = Tv.AgentMaxFn.apply(xd)
    md.backward(d.cuda())
    assert torch.equal(md.detach().cpu(), mr.detach())
    first = torch.zeros_like(x)
    idx = x.argmax(0)                                  # torch.argmax: the first occurrence
    first.scatter_(0, idx.unsqueeze(0), d)
    assert torch.equal(xd.grad.cpu(), first)


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_v2vnet(rng, agg=str(fx["agg"]))
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.v2vnet_param_spec(args), seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"])
            for i in range(len(types))]
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    dd["img_pairwise_t_matrix_collab"] = synth.v2vnet_pairwise(len(types), args["max_cav_num"])
    H, W = (int(v) for v in fx["head_hw"])
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    tgt = {k: torch.from_numpy(lc[k]).cuda() for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    return hy, args, sd, dd, tgt


def _model(args, sd):
    from airv2x_perception_amd.opencood_iface import Airv2xV2VNet
    m = Airv2xV2VNet(args)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


@pytest.mark.parametrize("name", ["train_v2vnet_small_n3", "train_v2vnet_small_n2_max", "train_v2vnet_full_n3"])
def test_v2vnet_training_step_matches_the_reference(name):
    fx = load_fixture(name)
    hy, args, sd, dd, tgt = _case(fx)
    model = _model(args, sd)
    out = model(dd)
    for k in ("psm", "rm", "obj"):
        assert out[k].requires_grad
        hs = int(fx["head_stride"])
        assert_close(out[k].detach().cpu()[..., ::hs, ::hs], fx[k], 3e-4, 3e-4 * float(np.abs(fx[k]).max()), k)
    # comm_rates (v2v_fuse.py:138,172) is kept in train mode too: non-zeros of the node features per (node, iteration) / B -- ReLU zeros, so
    # only elements within rounding of zero can differ from the reference's count
    assert isinstance(out["comm_rate"], float) and abs(out["comm_rate"] - float(fx["comm_rate"])) <= 1e-3 * float(fx["comm_rate"]), (out["comm_rate"], float(fx["comm_rate"]))
    total = _loss(args)(out, tgt)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - fx["losses"][0]) < 3e-4 * abs(fx["losses"][0])
    P = dict(model.named_parameters())
    keys = [str(k) for k in fx["grad_keys"]]
    have = sorted(k for k, p in P.items() if p.grad is not None)
    assert set(keys) <= set(have), sorted(set(keys) - set(have))
    C = args["v2vfusion"]["in_channels"]
    cell = "fusion_net.conv_gru.cell_list.0."
    # one step from a zero hidden state: the reset-gate rows and the hidden-state input columns carry no gradient -- exactly zero here
    # (no path), exactly zero in the reference (multiplied by the zero state)
    gw, cw = P[cell + "conv_gates.weight"].grad, P[cell + "conv_can.weight"].grad
    assert float(gw[:C].abs().max()) == 0.0 and float(gw[:, 2 * C:].abs().max()) == 0.0 and float(cw[:, 2 * C:].abs().max()) == 0.0
    assert float(P[cell + "conv_gates.bias"].grad[:C].abs().max()) == 0.0
    dev, refdev = {}, {}
    for k in keys:
        g = P[k].grad.reshape(-1)
        stride = max(1, g.numel() // 4096)
        gmax = float(fx["g64max:" + k])
        dev[k] = np.abs(g[::stride].cpu().numpy().astype(np.float64) - fx["g64:" + k].astype(np.float64)).max() / max(gmax, 1e-300)
        refdev[k] = float(fx["gdev:" + k])
    med_ref, med_dev = float(np.median(list(refdev.values()))), float(np.median(list(dev.values())))
    print(f"{name}: gradient deviation from float64, rel. to max -- device median {med_dev:.2e} worst {max(dev.values()):.2e}; "
          f"reference fp32 median {med_ref:.2e} worst {max(refdev.values()):.2e}")
    bad = {k: (dev[k], refdev[k]) for k in keys if dev[k] > 3.0 * refdev[k] + 2.0 * med_ref + 1e-4}
    assert not bad, bad
    assert med_dev <= 1.5 * med_ref + 1e-4, (med_dev, med_ref)
    assert max(dev.values()) <= 2.5 * max(refdev.values()) + 1e-4, (max(dev.values()), max(refdev.values()))
    for k, b in model.named_buffers():
        ref = fx["b:" + k].astype(np.float64)
        assert np.abs(b.detach().cpu().numpy().astype(np.float64) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k


def test_v2vnet_optimizer_steps_and_eval():
    fx = load_fixture("train_v2vnet_small_n3")
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
        o2 = v2v.v2vnet_forward(dd, sd_now, args)
    for k in ("psm", "rm", "obj"):
        assert_close(o1[k].cpu(), o2[k], 1e-3, 1e-3 * float(o2[k].abs().max()), k)
