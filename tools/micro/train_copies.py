"""Which python lines issue the torch copy / fill / cat kernels of one Where2Comm training step (torch.profiler with stacks)."""
import os, sys, random
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.airv2x_where2com import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
from oracle import voxelize_oracle as vox
dev = torch.device("cuda", 0)
hy = synth.default_hypes(None); args = hy["model"]["args"]; rng = synth.DEFAULT_RANGE; pp = hy["preprocess"]
types = synth.sort_types(synth.agent_types_for(4))[1]
voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 8192, rng), pp["cav_lidar_range"]), pp["cav_lidar_range"],
                             pp["args"]["voxel_size"], 32, 32000) for i in range(4)]
dd = synth.data_dict_to(synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"]), dev)
m = Airv2xWhere2com(args); m.load_state_dict(synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)); m = m.to(dev).train()
m.sync_comm_rate = False
lc = synth.loss_case(100, B=1, H=100, W=352, A=2, C=7, pos_frac=0.002)
tgt = {k: torch.from_numpy(lc[k]).to(dev) for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
crit = PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": 7})
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
random.seed(0)


def step():
    opt.zero_grad(set_to_none=True)
    out = m(dd)
    crit(out, tgt).backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = {}
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith("CPU") and ev.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::add_", "aten::add", "aten::mul"):
        st = [s for s in (ev.stack or []) if "airv2x_perception_amd" in s or "train_host" in s or "train_copies" in s]
        key = (ev.name, st[0] if st else "(autograd engine / torch internals)")
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:4d}  {v[1]:9.1f} us  {k[0]:12s} {k[1]}")
