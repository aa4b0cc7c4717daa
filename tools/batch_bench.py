#!/usr/bin/env python3
"""Throughput of B frames batched into ONE forward (reference collate layout) vs frames in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = [sys.argv[0], "--cpu-frames", "0"]
a = bench.parse()
torch.set_num_threads(4)
hy, args, dd, clouds, types = bench.build_inputs(a.agents, a.points, torch.device("cuda"), model="where2com")
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
m = Airv2xWhere2com(args); m.load_state_dict(synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)); m = m.cuda().eval()
m.sync_comm_rate = False
for B in (1, 2, 3, 4):
    batch = synth.merge_frames([dd] * B) if B > 1 else dd
    for sk in (True, False):
        m.engine().stream_k = sk
        for _ in range(3): m(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 20
        for _ in range(N): m(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        print(f"B={B} stream_k={sk}: {dt * 1e3:.2f} ms per forward, {B / dt:.1f} frames/s", flush=True)
