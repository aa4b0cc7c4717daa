"""Soak: 3 000 frames through a 3-deep FramePipeline, two different clouds alternating in runs of seven, every frame compared bit for bit
(heads + comm_rate) with the single-stream frame of the same engine mode -- stale occupancy bytes, counters or workspace races would show.
Run on a GPU box: python tools/soak_pipeline.py (last run: 0 mismatches at 4 and 8 agents)."""
import sys, torch
sys.path.insert(0, ".")
from types import SimpleNamespace
import bench
from airv2x_perception_amd.opencood_iface.engine import FramePipeline
dev = torch.device("cuda", 0)
for agents in (4, 8):
    a = SimpleNamespace(model="where2com", amp=False, gemm="x3", agents=agents, points=8192, mods=("lidar",))
    hy, args, dd, clouds, types = bench.build_inputs(agents, 8192, dev, only=None, model="where2com", modalities=("lidar",))
    hy2, args2, dd2, _, _ = bench.build_inputs(agents, 6000, dev, only=None, model="where2com", modalities=("lidar",))
    model, eng, sd = bench.make_model(a, args, dev)
    eng.throughput_mode = True
    refs = []
    for d in (dd, dd2):
        o = model(d); torch.cuda.synchronize()
        refs.append({k: o[k].clone() for k in ("psm", "rm", "obj")} | {"comm_rate": o["comm_rate"].clone() if torch.is_tensor(o["comm_rate"]) else o["comm_rate"]})
    pipe = FramePipeline(eng, 3)
    pend, bad, N = [], 0, 3000
    for f in range(N):
        w = (f // 7) % 2          # alternate the two clouds in runs of 7 (stale occupancy / counters would show)
        pend.append((w,) + pipe.submit(dd if w == 0 else dd2))
        if len(pend) == 3:
            w0, o, ev = pend.pop(0); ev.synchronize()
            ok = all(torch.equal(o[k], refs[w0][k]) for k in ("psm", "rm", "obj")) and int(o["comm_rate"]) == int(refs[w0]["comm_rate"])
            bad += (not ok)
    print(f"agents {agents}: {N} pipelined frames alternating two clouds, mismatches: {bad}")
