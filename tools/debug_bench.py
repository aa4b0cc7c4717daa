import faulthandler, sys, os, time
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.voxelizer import voxelize_points
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
def log(*a):
    print(*a, flush=True)
hy = synth.default_hypes(); pp = hy["preprocess"]; args = hy["model"]["args"]
pts = torch.from_numpy(synth.synthetic_cloud(0, 8192)).cuda()
log("voxelize...")
v = voxelize_points(pts, pp["cav_lidar_range"], pp["args"]["voxel_size"], 32, 70000, range_filter=True)
log("M =", v[0].shape)
types = ["vehicle", "vehicle", "rsu", "drone"]
voxd = [voxelize_points(torch.from_numpy(synth.synthetic_cloud(i, 8192)).cuda(), pp["cav_lidar_range"], pp["args"]["voxel_size"], 32, 70000, range_filter=True) for i in range(4)]
dd = synth.build_data_dict_device(voxd, types, "cuda")
sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), 0)
m = Airv2xWhere2com(args); m.load_state_dict(sd); m = m.cuda().eval(); m.sync_comm_rate = False
log("eager forward...")
o = m(dd); torch.cuda.synchronize(); log("eager ok", float(o["psm"].abs().max()))
t0 = time.perf_counter()
for _ in range(10): o = m(dd)
torch.cuda.synchronize(); log("eager ms/frame", (time.perf_counter() - t0) * 100)
m.engine().use_graph = True
log("graph capture...")
o2 = m(dd); torch.cuda.synchronize(); log("graph ok", float((o2["psm"] - o["psm"]).abs().max()))
t0 = time.perf_counter()
for _ in range(10): o2 = m(dd)
torch.cuda.synchronize(); log("graph ms/frame", (time.perf_counter() - t0) * 100)
