#!/usr/bin/env python3
"""Frames batched into ONE forward (the reference's collate layout, B samples) against / together with frames in flight on separate
streams, today's kernels (x3 mode, throughput-mode Winograd classes in every leg so that only the SCHEDULE differs).
VERDICT r05 item 1b: B = 3 turns the 60-144-workgroup launches of the 4-agent frame into 180-432, amortises launch + drain and the
once-per-XCD weight-plane fetch.  Prints frames/s per (B, frames in flight); profiles/r06_batch_vs_inflight.txt holds the last run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.engine import FramePipeline

agents = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
a = SimpleNamespace(model="where2com", amp=False, gemm="x3", agents=agents, points=8192, mods=("lidar",))
hy, args, dd, clouds, types = bench.build_inputs(agents, 8192, dev, only=None, model="where2com", modalities=("lidar",))
model, eng, sd = bench.make_model(a, args, dev)
N = 60
for B, depth in ((1, 1), (1, 3), (2, 1), (3, 1), (4, 1), (6, 1), (2, 2), (3, 2), (2, 3), (3, 3)):
    batch = synth.merge_frames([dd] * B) if B > 1 else dd
    batch = synth.data_dict_to(batch, dev)
    pipe = FramePipeline(eng, depth)
    pipe.throughput_mode = True
    for _ in range(2 * depth + 1):
        pipe.submit(batch)
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        pipe.submit(batch)
    pipe.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"agents {agents}  B={B} frames per forward x {depth} in flight: {dt * 1e3:.3f} ms per forward, {B / dt:.1f} frames/s", flush=True)
