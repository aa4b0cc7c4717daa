"""Host-side profile of one camera + LiDAR frame (8 agents): where the python time and the copy calls come from."""
import cProfile, os, pstats, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
a = SimpleNamespace(model="where2com", amp=False, gemm="x3", agents=8, points=8192, mods=("cam", "lidar"))
hy, args, dd, clouds, types = bench.build_inputs(8, 8192, dev, only=None, model="where2com", modalities=("cam", "lidar"))
model, eng, sd = bench.make_model(a, args, dev)
for _ in range(3):
    model(dd)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    model(dd)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / 5:.2f} ms per frame, synchronised {1e3 * (t2 - t0) / 5:.2f} ms per frame")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    model(dd)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
