"""Throughput of (a) agent groups on 1/2/3 streams inside one frame, (b) two frames in flight."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com

nag = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hy, args, dd, clouds, types = bench.build_inputs(nag, 8192, torch.device("cuda"))
sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
m = Airv2xWhere2com(args); m.load_state_dict(sd); m = m.cuda().eval(); m.sync_comm_rate = False
eng = m.engine()
ref = None
for ns in (1, 2, 3, 4):
    eng.agent_streams = ns
    o = m(dd); o = m(dd)
    torch.cuda.synchronize()
    if ref is None:
        ref = o
    same = all(torch.equal(o[k], ref[k]) for k in ("psm", "rm", "obj")) and int(o["comm_rate"]) == int(ref["comm_rate"]) and float(o["com"]) == float(ref["com"])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(40):
            o = m(dd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
    print(f"agents {nag} agent_streams {ns}: {dt*1e3:.3f} ms/frame {1/dt:.1f} fps  bit-identical={same}", flush=True)
