#!/usr/bin/env python3
"""Emulated latency of the agent-sharded frame (one agent per GPU) on ONE GPU: time the per-rank local stage with one
agent and the ego stage on the buffer `world` ranks would have gathered (the local send buffer repeated `world` times --
the ego stage's cost does not depend on the values).  The all-gather itself is not emulated; its size is printed.
Usage: python tools/shard_latency.py [--model where2com|cobevt|v2xvit|when2com] [--world 8] [--steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from airv2x_perception_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="where2com")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--graph", action="store_true", help="engine.use_graph: replay the two stages from hipGraphs (Where2Comm)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from airv2x_perception_amd import opencood_iface as oi
    hy, args, dd, clouds, types = bench.build_inputs(a.world, 8192, dev, only=[0], model=a.model)   # rank 0: the ego only
    spec, M = {"where2com": (synth.where2com_param_spec, oi.Airv2xWhere2com), "cobevt": (synth.cobevt_param_spec, oi.Airv2xCoBEVT),
               "v2xvit": (synth.v2xvit_param_spec, oi.Airv2xV2XVit), "when2com": (synth.when2com_param_spec, oi.Airv2xWhen2com)}[a.model]
    model = M(args)
    model.load_state_dict(synth.synthetic_state_dict(spec(args), seed=0))
    model = model.to(dev).eval()
    eng = model.engine()
    eng.use_graph = a.graph
    dd["shard_rank"] = 0
    rbuf = {}

    def gathered(send):        # the buffer `world` ranks would have gathered, at a STABLE address (as EngineBackend.recv_buffer gives the real frame)
        if "t" not in rbuf:
            rbuf["t"] = torch.empty(a.world * send.numel(), dtype=send.dtype, device=send.device)
        rbuf["t"].view(a.world, -1).copy_(send.view(1, -1).expand(a.world, -1))
        return rbuf["t"]

    def frame():
        send, stats, meta = eng.shard_local_stage(dd, has_ego=True)
        recv = gathered(send)
        return eng.shard_ego_stage(recv, stats, meta, a.world), send

    for _ in range(3):
        out, send = frame()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tl = te = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ev[0].record()
        send, stats, meta = eng.shard_local_stage(dd, has_ego=True)
        ev[1].record()
        recv = gathered(send)
        ev[2].record()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.shard_ego_stage(recv, stats, meta, a.world)
        e1.record()
        torch.cuda.synchronize()
        tl += ev[0].elapsed_time(ev[1])
        te += e0.elapsed_time(e1)
    tl, te = tl / a.steps, te / a.steps
    # wall clock of whole frames back to back (host launch cost included: what a rank of an 8-GPU group pays per frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        frame()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"{a.model}: world={a.world}  {'hipGraph replay' if a.graph else 'eager launches'}: local + copy + ego stage, back to back: {wall:.3f} ms wall per frame")
    tp = None
    if hasattr(eng, "shard_ego_partial"):      # second level: this rank's share of the fusion + the gather of the head outputs
        if hasattr(eng, "gap_exchange"):       # V2X-ViT: the per-block (n, C) all-reduce is not run here (1-GPU box)
            eng.gap_exchange = lambda g, w, i: None
        send, stats, meta = eng.shard_local_stage(dd, has_ego=True)
        recv = send.repeat(a.world)
        for _ in range(2):
            part, ctx = eng.shard_ego_partial(recv, stats, meta, a.world, 0)
            eng.shard_ego_finish(part.repeat(a.world), ctx, a.world)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            part, ctx = eng.shard_ego_partial(recv, stats, meta, a.world, 0)
            eng.shard_ego_finish(part.repeat(a.world), ctx, a.world)
        e1.record()
        torch.cuda.synchronize()
        tp = e0.elapsed_time(e1) / a.steps
    mb = send.numel() * 4 / 1e6
    link = 153.0   # GB/s per xGMI link (guide); 7 peers write into each GPU over separate links
    comm = mb / 1e3 / link * 1e3
    print(f"{a.model}: world={a.world}  local stage (1 agent) {tl:.3f} ms | message {mb:.2f} MB per agent, all-gather >= {comm:.3f} ms at "
          f"{link:.0f} GB/s per link | ego stage ({a.world} agents) {te:.3f} ms | frame >= {tl + comm + te:.3f} ms -> <= {1e3 / (tl + comm + te):.1f} frames/s per "
          f"{a.world}-GPU group (strictly sequential frames; GPU time only)")
    if tp is not None:
        print(f"{a.model}: two-level (fusion split over the ranks by columns): partial fusion + heads + finish {tp:.3f} ms instead of {te:.3f} ms "
              f"-> frame >= {tl + comm + tp:.3f} ms -> <= {1e3 / (tl + comm + tp):.1f} frames/s per {a.world}-GPU group")


if __name__ == "__main__":
    main()
