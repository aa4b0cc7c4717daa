#!/usr/bin/env python3
"""Agent-sharded frame (incl. the second-level sharding of the CoBEVT / V2X-ViT fusion) with REAL processes and
collectives, against the single-process forward of the same frame.  On a multi-GPU node:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/shard_check.py --model v2xvit
On a one-GPU box both ranks can share the device (RCCL refuses that, gloo stages through the host):
    AV2X_ONE_DEVICE=1 AV2X_DIST_BACKEND=gloo python -m torch.distributed.run ... tools/shard_check.py --model cobevt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from airv2x_perception_amd import opencood_iface as oi
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.sharded import EngineBackend, ShardedFrame, partition_agents


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="where2com", choices=["where2com", "cobevt", "v2xvit", "when2com"])
    ap.add_argument("--agents", type=int, default=4)
    a = ap.parse_args()
    local = 0 if os.environ.get("AV2X_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    backend = os.environ.get("AV2X_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", local)
    mine = list(partition_agents(a.agents, world)[rank])
    hy, args, dd, _, _ = bench.build_inputs(a.agents, 8192, dev, only=mine, model=a.model)
    _, _, dd_full, _, _ = bench.build_inputs(a.agents, 8192, dev, only=None, model=a.model)
    spec, M = {"where2com": (synth.where2com_param_spec, oi.Airv2xWhere2com), "cobevt": (synth.cobevt_param_spec, oi.Airv2xCoBEVT),
               "v2xvit": (synth.v2xvit_param_spec, oi.Airv2xV2XVit), "when2com": (synth.when2com_param_spec, oi.Airv2xWhen2com)}[a.model]
    model = M(args)
    model.load_state_dict(synth.synthetic_state_dict(spec(args), seed=0))
    model = model.to(dev).eval()
    eng = model.engine()
    eng.stream_k = False
    out = ShardedFrame(EngineBackend(eng)).forward(dd)
    ref = eng.forward(dd_full)
    err = max(float((out[k] - ref[k]).abs().max()) for k in ("psm", "rm", "obj"))
    mag = max(float(ref[k].abs().max()) for k in ("psm", "rm", "obj"))
    print(f"rank {rank}/{world} {a.model}: sharded vs single-process max |diff| {err:.3e} (max |ref| {mag:.2f})", flush=True)
    assert err <= 1e-4 * max(1.0, mag), "sharded frame differs from the single-process forward"
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
