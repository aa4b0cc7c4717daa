#!/usr/bin/env python3
"""Print the autotuned conv schedule of a model (tile, flags, workgroups) per layer shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
a = bench.parse()
hy, args, dd, clouds, types = bench.build_inputs(a.agents, a.points, torch.device("cuda"), model=a.model)
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT, Airv2xV2XVit, Airv2xWhere2com
cls, spec = {"cobevt": (Airv2xCoBEVT, synth.cobevt_param_spec), "v2xvit": (Airv2xV2XVit, synth.v2xvit_param_spec),
             "where2com": (Airv2xWhere2com, synth.where2com_param_spec)}[a.model]
m = cls(args); m.load_state_dict(synth.synthetic_state_dict(spec(args), seed=0)); m = m.cuda().eval()
m(dd); torch.cuda.synchronize()
for k, (t, g) in sorted(m.engine().tile_cache.items(), key=lambda kv: -kv[0][1]):
    bm, bn = t >> 16, t & 0xfff
    print(f"mode={k[0]} M={k[1]:7d} cin={k[2]:4d} coutp={k[3]:5d} ks={k[4]} s={k[5]} sk_ok={k[6]} -> {bm}x{bn}"
          f"{'w8' if t & 0x8000 else ''}{'d' if t & 0x4000 else ''}{'sk' if t & 0x2000 else ''}{'p' if t & 0x1000 else ''} wgs={g}")
