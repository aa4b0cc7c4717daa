"""Which python lines issue copies in one camera + LiDAR frame (torch.profiler, python stacks)."""
import os, sys
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda", 0)
a = SimpleNamespace(model="where2com", amp=False, gemm="x3", agents=8, points=8192, mods=("cam", "lidar"))
hy, args, dd, clouds, types = bench.build_inputs(8, 8192, dev, only=None, model="where2com", modalities=("cam", "lidar"))
model, eng, sd = bench.make_model(a, args, dev)
for _ in range(3):
    model(dd)
torch.cuda.synchronize()
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=cfg) as prof:
    model(dd)
    torch.cuda.synchronize()
agg = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_"):
        st = [s for s in (ev.stack or []) if "airv2x_perception_amd" in s or "bench.py" in s]
        key = (ev.name, st[0] if st else "(no python frame)")
        agg[key] = agg.get(key, 0) + 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v:5d}  {k[0]:14s} {k[1]}")
