"""Ad-hoc timing of the post-process on real model outputs (debug tool)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor

hy, args, dd, clouds, types = bench.build_inputs(4, 8192, torch.device("cuda"))
sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
model = Airv2xWhere2com(args); model.load_state_dict(sd); model = model.cuda().eval(); model.sync_comm_rate = False
model.engine().use_graph = True
post = VoxelPostprocessor(hy["postprocess"])
data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(post.generate_anchor_box())}}
for _ in range(3):
    o = model(dd)
torch.cuda.synchronize()


def T(f, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


print("model ms", T(lambda: model(dd))[0])
ms, r = T(lambda: post.post_process_airv2x(data, {"ego": o}, return_counts=True))
print("post ms", ms, r[4])
print("model+post ms", T(lambda: post.post_process_airv2x(data, {"ego": model(dd)}, return_counts=True))[0])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); post.post_process_airv2x(data, {"ego": o}); ev[1].record(); torch.cuda.synchronize()
print("post gpu ms (events)", ev[0].elapsed_time(ev[1]))
print("model+sync ms (graph)", T(lambda: (model(dd), torch.cuda.synchronize()))[0])
model.engine().use_graph = False
print("model ms (eager)", T(lambda: model(dd))[0])
print("model+sync ms (eager)", T(lambda: (model(dd), torch.cuda.synchronize()))[0])
print("model+post ms (eager)", T(lambda: post.post_process_airv2x(data, {"ego": model(dd)}, return_counts=True))[0])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): r = post.post_process_airv2x(data, {"ego": model(dd)}, return_counts=True)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
