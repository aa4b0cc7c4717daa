"""Per-tensor gradient deviation of the device training step against a train_* fixture (diagnostic)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.helpers import load_fixture, train_case_from_fixture
from airv2x_perception_amd.opencood_iface.airv2x_where2com import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
from airv2x_perception_amd.opencood_iface.train_where2com import forward_train
name = sys.argv[1] if len(sys.argv) > 1 else "train_full_n4"
fx = load_fixture(name)
hy, args, sd, dd, tgt = train_case_from_fixture(fx)
if len(sys.argv) > 2 and sys.argv[2] == "direct":
    os.environ["AV2X_WINOGRAD"] = "0"
m = Airv2xWhere2com(args); m.load_state_dict(sd); m = m.cuda().train()
n, _, H, W = [int(v) for v in fx["mask_shape"]]
ref_mask = torch.from_numpy(np.unpackbits(fx["mask"])[: n * H * W].reshape(n, H, W).astype(np.float32))
out = forward_train(m, dd, topk=[int(k) for k in fx["K"]], mask=ref_mask)
hs = int(fx["head_stride"])
for k in ("psm", "rm", "obj"):
    print(k, float(np.abs(out[k].detach()[..., ::hs, ::hs].cpu().numpy() - fx[k]).max()), float(np.abs(fx[k]).max()))
crit = PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})
total = crit(out, {k: v.cuda() for k, v in tgt.items()})
total.backward()
print("loss", float(total.detach()), fx["losses"][0])
P = dict(m.named_parameters())
for k in [str(k) for k in fx["grad_keys"]]:
    g = P[k].grad.reshape(-1)
    stride = max(1, g.numel() // 4096)
    gmax = fx["gsum:" + k][2]
    err = np.abs(g[::stride].cpu().numpy().astype(np.float64) - fx["g:" + k]).max()
    asum = g.double().abs().sum().item()
    print(f"{err / gmax:10.3e}  asum {asum / fx['gsum:' + k][1] - 1:+.2e}  max {gmax:9.3e}  {k}")
