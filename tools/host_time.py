#!/usr/bin/env python3
"""Host (CPU) time per frame submission of the Where2Comm pipeline: enqueue cost vs GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = [sys.argv[0], "--cpu-frames", "0"]
a = bench.parse()
torch.set_num_threads(2)
hy, args, dd, clouds, types = bench.build_inputs(a.agents if a.agents > 0 else 4, a.points, torch.device("cuda"), model="where2com")
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.engine import FramePipeline
m = Airv2xWhere2com(args); m.load_state_dict(synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)); m = m.cuda().eval()
m.sync_comm_rate = False
m(dd); torch.cuda.synchronize()
pipe = FramePipeline(m.engine(), 3)
for _ in range(6): pipe.submit(dd)
pipe.drain(); torch.cuda.synchronize()
for use_graph in (False, True):
    for e in pipe.engines: e.use_graph = use_graph
    for _ in range(6): pipe.submit(dd)
    pipe.drain(); torch.cuda.synchronize()
    N = 60
    t0 = time.perf_counter(); cpu = 0.0
    for _ in range(N):
        c0 = time.process_time(); pipe.submit(dd); cpu += time.process_time() - c0
    t_enq = time.perf_counter() - t0
    pipe.drain(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"use_graph={use_graph}: enqueue wall {t_enq / N * 1e3:.2f} ms/frame, CPU {cpu / N * 1e3:.2f} ms/frame, total {t_all / N * 1e3:.2f} ms/frame ({N / t_all:.1f} fps)")
