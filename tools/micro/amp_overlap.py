"""AMP training step with the weight gradients on the side stream (overlapping the main stream) vs on the main stream: the parameter gradients
must be the same bits -- the side stream's bf16-MFMA kernels run beside ATen / library kernels this build cannot lint.
python tools/micro/amp_overlap.py [where2com|cobevt|v2xvit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import random                                                                       # noqa: E402

from airv2x_perception_amd import opencood_iface as oi                              # noqa: E402
from airv2x_perception_amd import synth                                             # noqa: E402
from airv2x_perception_amd.opencood_iface import train_ops as T                     # noqa: E402
from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass     # noqa: E402
from oracle import voxelize_oracle as vox                                           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "where2com"
Model, hypes_fn, spec_fn = {"where2com": (oi.Airv2xWhere2com, synth.default_hypes, synth.where2com_param_spec),
                            "cobevt": (oi.Airv2xCoBEVT, synth.default_hypes_cobevt, synth.cobevt_param_spec),
                            "v2xvit": (oi.Airv2xV2XVit, synth.default_hypes_v2xvit, synth.v2xvit_param_spec)}[name]
dev = torch.device("cuda", 0)
hy = hypes_fn(None)
args = hy["model"]["args"]
if name == "cobevt":
    args["fax_fusion"]["drop_out"] = 0.0
rng, pp = synth.DEFAULT_RANGE, hy["preprocess"]
types = synth.sort_types(synth.agent_types_for(4))[1]
voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 8192, rng), pp["cav_lidar_range"]), pp["cav_lidar_range"],
                             pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"]) for i in range(4)]
dd_host = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
if name == "v2xvit":
    import numpy as np
    g_ = np.random.default_rng(99)
    scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
    for i in range(1, 4):
        scm[0, i] = torch.from_numpy(synth.se2_correction(g_.uniform(-10, 10), g_.uniform(-8, 8), g_.uniform(-8, 8)))
    dd_host["spatial_correction_matrix"] = scm
dd = synth.data_dict_to(dd_host, dev)
sd = synth.synthetic_state_dict(spec_fn(args), seed=0)
g = [int(v) for v in args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]]
lc = synth.loss_case(100, B=1, H=g[1] // 2, W=g[0] // 2, A=args["anchor_number"], C=args["num_class"], pos_frac=0.002)
tgt = {k: torch.from_numpy(lc[k]).to(dev) for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
crit = PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})


def grads(overlap):
    T.OVERLAP_WGRAD = overlap
    torch.manual_seed(0)
    random.seed(0)
    m = Model(args)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    m.sync_comm_rate = False
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(m(dd), tgt)
    loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, float(loss.detach())


ref, l0 = grads(False)
bad_total = 0
for rep in range(int(os.environ.get("REPS", "5"))):
    got, l1 = grads(True)
    bad = [k for k in ref if not torch.equal(ref[k], got[k])]
    bad_total += len(bad)
    print(f"{name} rep {rep}: loss {l1:.6f} (serial {l0:.6f}); {len(bad)} of {len(ref)} gradients differ between overlapped and serial weight gradients", bad[:4], flush=True)
print("TOTAL differing:", bad_total)
