#!/usr/bin/env python3
"""Measured margin of the V2X-ViT head outputs against the reference's golden (tests/golden/v2xvit_*): max |hip - ref| relative to the
mixed bound rtol |ref| + atol for a sweep of (rtol, atol) -- the number DESIGN.md section 4 quotes for the per-model tolerance.
Usage (GPU): python tools/v2xvit_margin.py [fixture ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tests.test_v2xvit as tv
from tests.helpers import load_fixture
from airv2x_perception_amd.opencood_iface import Airv2xV2XVit

for name in (sys.argv[1:] or ["v2xvit_small_n3", "v2xvit_full_n4", "v2xvit_full_n8"]):
    fx = load_fixture(name)
    hy, args, sd, dd = tv._case(fx)
    model = Airv2xV2XVit(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    for mode in ("x3", "f32"):
        eng = model.engine()
        eng.wino_x3 = eng.x3p = eng.wino4_x3 = mode == "x3"
        out = model(dd)
        hs = int(fx["head_stride"]) if "head_stride" in fx else 1
        line = []
        for k in ("psm", "rm", "obj"):
            got = out[k].cpu().numpy()[..., ::hs, ::hs].astype(np.float64)
            ref = fx[k].astype(np.float64)
            err = np.abs(got - ref)
            mx = float(np.abs(ref).max())
            worst = {f"{rt:g}": float((err / (rt * np.abs(ref) + rt * 0.1 * max(10.0, mx))).max()) for rt in (1e-3, 5e-4, 2e-4)}
            line.append(f"{k}: max|err| {err.max():.2e} (max|ref| {mx:.1f}, {err.max() / mx:.2e} of it) fraction of bound at rtol {worst}")
        print(f"{name} [{mode}]  " + " | ".join(line), flush=True)
