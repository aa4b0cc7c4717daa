"""Host-side cost of issuing one training step (time for the python calls to RETURN, the GPU running behind) vs the
synchronised step time: tells whether the step is host-bound."""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
import torch
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
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = m(dd); t1 = time.perf_counter()
    loss = crit(out, tgt); loss.backward(); t2 = time.perf_counter()
    opt.step(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 3:
        print(f"host: forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms, opt {1e3*(t3-t2):.2f} ms; synchronised step {1e3*(t4-t0):.2f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        pr.enable()
        out = m(dd)
        pr.disable()
        crit(out, tgt).backward(); opt.step()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
