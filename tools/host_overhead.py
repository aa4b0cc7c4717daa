import time, torch, sys
sys.path.insert(0, '/root/repo')
from airv2x_perception_amd import synth
import bench
a = bench.parse(["--agents", "4"])
dev = torch.device("cuda", 0)
# tiny grid: GPU work is negligible, the wall time per frame is the host-side cost of issuing the frame
rng = [-12.8, -6.4, -3, 12.8, 6.4, 1]
hy = synth.default_hypes(rng)
args = hy["model"]["args"]
from oracle import voxelize_oracle as vox
types = synth.sort_types(synth.agent_types_for(4))[1]
voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 400, rng), rng), rng, [0.4, 0.4, 4.0]) for i in range(4)]
dd = synth.data_dict_to(synth.build_data_dict(voxd, types), dev)
model, eng, sd = bench.make_model(a, args, dev)
for _ in range(5): model(dd)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(200): model(dd)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/200
print(f"eager: {dt*1e3:.3f} ms per frame on a {args['anchor_number']}-anchor tiny grid (host-bound)")
eng.use_graph=True
for _ in range(3): model(dd)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(200): model(dd)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/200
print(f"graph: {dt*1e3:.3f} ms per frame")
