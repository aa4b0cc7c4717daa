"""One training step of Airv2xWhere2com / Airv2xCoBEVT / Airv2xV2XVit / Airv2xWhen2com / Airv2xV2VNet (--model) on the default AirV2X grid (704 x 200 canvas, N agents x 8192 points), on the device:
train-mode forward (BatchNorm batch statistics, random top-K mask), PointPillarLossMultiClass, backward, Adam step.
Prints one JSON line: ms per step (forward / loss+backward / optimiser), peak memory, and -- with --cpu -- the oracle's
(torch CPU autograd) time for the same step.  Not the headline metric (BASELINE.json's is inference frames/s)."""
import argparse
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(agents=4, steps=10, warmup=3, small=False, cpu=False, dev=None, dd=None, args=None, model_name="where2com", amp=False, modalities=None):
    """-> dict (see the module docstring).  ``dd`` / ``args``: a frame already on the device and its model args (bench.py
    passes the one its device voxelizer built); otherwise the frame is built here with the oracle's CPU voxelizer."""
    from types import SimpleNamespace
    a = SimpleNamespace(agents=agents, steps=steps, warmup=warmup, small=small, cpu=cpu)
    import numpy as np
    from airv2x_perception_amd import opencood_iface as oi
    from airv2x_perception_amd import synth
    from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass
    Model, hypes_fn, spec_fn = {"where2com": (oi.Airv2xWhere2com, synth.default_hypes, synth.where2com_param_spec),
                                "cobevt": (oi.Airv2xCoBEVT, synth.default_hypes_cobevt, synth.cobevt_param_spec),
                                "v2xvit": (oi.Airv2xV2XVit, synth.default_hypes_v2xvit, synth.v2xvit_param_spec),
                                "when2com": (oi.Airv2xWhen2com, synth.default_hypes_when2com, synth.when2com_param_spec),
                                "v2vnet": (oi.Airv2xV2VNet, synth.default_hypes_v2vnet, synth.v2vnet_param_spec)}[model_name]
    dev = dev or torch.device("cuda", 0)
    dd_host = None
    if dd is None:
        from oracle import voxelize_oracle as vox
        rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0] if a.small else None
        if modalities:           # camera branch (LSS encoder per agent type, gt-depth lift): Airv2xWhere2com only
            assert model_name == "where2com" and not amp
            fd = (104, 168) if a.small else (360, 640)
            hy = synth.multimodal_hypes(tuple(modalities), rng, fd, True)
        else:
            hy = hypes_fn(rng)
        args = hy["model"]["args"]
        rng = rng or synth.DEFAULT_RANGE
        types = synth.sort_types(synth.agent_types_for(a.agents))[1]
        pp = hy["preprocess"]
        voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 700 if a.small else 8192, rng), pp["cav_lidar_range"]),
                                     pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"],
                                     pp["args"]["max_voxel_train"]) for i in range(a.agents)]
        dd_host = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
        if model_name == "v2xvit":    # seeded SE(2) correction per non-ego agent + the frame's prior encoding (bench.py build_inputs)
            g_ = np.random.default_rng(99)
            scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
            for i in range(1, a.agents):
                scm[0, i] = torch.from_numpy(synth.se2_correction(g_.uniform(-10, 10), g_.uniform(-8, 8), g_.uniform(-8, 8)))
            dd_host["spatial_correction_matrix"] = scm
        if model_name == "when2com":
            dd_host["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(a.agents, args["max_cav_num"])
        if model_name == "v2vnet":
            dd_host["img_pairwise_t_matrix_collab"] = synth.v2vnet_pairwise(a.agents, args["max_cav_num"])
        if modalities:
            dd_host = synth.add_cameras(dd_host, types, seed=50, final_dim=fd)
        dd = synth.data_dict_to(dd_host, dev)
    sd = synth.synthetic_state_dict(spec_fn(args), seed=0)
    model = Model(args)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    model.sync_comm_rate = False
    g = [int(v) for v in args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]]
    H, W = g[1] // 2, g[0] // 2
    if modalities:
        res_ = args["vehicle"]["cam"]["grid_conf"]["xbound"][2] / 0.4      # camera BEV cell over the lidar voxel
        H, W = int(round(g[1] / res_)) // 2, int(round(g[0] / res_)) // 2
    lc = synth.loss_case(100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=0.002)
    tgt_host = {k: torch.from_numpy(lc[k]) for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    tgt = {k: v.to(dev) for k, v in tgt_host.items()}
    crit = PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    random.seed(0)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0) if amp else None
    torch.cuda.reset_peak_memory_stats()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_f = t_b = t_o = 0.0
    losses = []
    for step in range(a.warmup + a.steps):
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        opt.zero_grad(set_to_none=True)
        e0.record()
        if amp:     # tools/train.py:50,107-130 of the reference: forward + loss under autocast, GradScaler around backward / step
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(dd)
                e1.record()
                loss = crit(out, tgt)
            scaler.scale(loss).backward()
            e2.record()
            scaler.step(opt)
            scaler.update()
        else:
            out = model(dd)
            e1.record()
            loss = crit(out, tgt)
            loss.backward()
            e2.record()
            opt.step()
        e3.record()
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        if step >= a.warmup:
            t_f += e0.elapsed_time(e1)
            t_b += e1.elapsed_time(e2)
            t_o += e2.elapsed_time(e3)
    k = a.steps
    res = {"what": f"{Model.__name__} training step (train-mode forward + PointPillarLossMultiClass + backward + Adam)" + (" under autocast(bf16) + GradScaler" if amp else ""),
           "agents": a.agents, "grid": [g[0], g[1]], "modalities": list(modalities) if modalities else ["lidar"], "steps": k, "ms_per_step": round((t_f + t_b + t_o) / k, 3),
           "ms_forward": round(t_f / k, 3), "ms_loss_backward": round(t_b / k, 3), "ms_optimizer": round(t_o / k, 3),
           "steps_per_s": round(1e3 * k / (t_f + t_b + t_o), 3), "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)], "dtype": "bf16 operands (autocast), fp32 master weights" if amp else "f32", "data": "synthetic"}
    # ---- the same K steps WITHOUT a host synchronisation per step (the loss dictionary is read back lazily, loss.py; a loop that logs every
    # N-th step): the host issues step i + 1 while the GPU finishes step i -- wall clock over K steps, a synchronise on both sides
    if not amp:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            opt.zero_grad(set_to_none=True)
            crit(model(dd), tgt).backward()
            opt.step()
        torch.cuda.synchronize()
        res["ms_per_step_no_host_sync"] = round(1e3 * (time.perf_counter() - t0) / a.steps, 3)
    # ---- roofline of the step's MFMA kernels: a second pass with an event pair around every convolution launch (forward and data
    # gradients go through the engine's launcher: its profile hook; weight gradients: train_ops' hook), weight gradients on the
    # main stream so that the pairs do not overlap
    from airv2x_perception_amd.opencood_iface import train_ops as T
    from airv2x_perception_amd.opencood_iface.autograd import _runner
    r = _runner(dev)
    overlap, T.OVERLAP_WGRAD = T.OVERLAP_WGRAD, False
    r.profile, r.wgrad_profile = [], []
    psteps = max(2, min(4, a.steps))
    for _ in range(psteps):
        opt.zero_grad(set_to_none=True)
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                l_ = crit(model(dd), tgt)
            scaler.scale(l_).backward()
        else:
            crit(model(dd), tgt).backward()
    torch.cuda.synchronize()
    prof, wprof = r.profile, r.wgrad_profile
    r.profile, r.wgrad_profile, T.OVERLAP_WGRAD = None, None, overlap
    PEAK = 2500.0 if amp else 157.3
    groups = {"conv_wino_f32 (forward + data gradients)": [0, 0.0, 0.0, 0.0], "conv_igemm_f32 (forward + data gradients, 1x1 / stride 2 / deconv)": [0, 0.0, 0.0, 0.0],
              "conv_wgrad (weight gradients)": [0, 0.0, 0.0, 0.0]}
    X3P = "conv_igemm_x3p (forward + data gradients, 1x1 / stride 2 / deconv / Linear: split-3 operands, six bf16 MFMA products per fp32 product)"
    W4X3 = "conv_wino4_x3 (forward + data gradients of the F(4x4,3x3) class: split-3 operands on the bf16 matrix cores)"
    gpeak = {X3P: 2500.0, W4X3: 2500.0}
    for tile, flops, e0, e1, wgs, shp in prof:
        wino = bool(tile[0] & 0x4000)
        f4 = wino and bool(tile[0] & 0x2000)                    # F(4x4,3x3): 36 products per 16 outputs = 1/4 of the direct count
        w4x3 = f4 and bool(tile[1] & 0x0400)
        x3p = not wino and not amp and bool(tile[1] & 0x1000)
        gk = W4X3 if w4x3 else "conv_wino_f32 (forward + data gradients)" if wino else (X3P if x3p else "conv_igemm_f32 (forward + data gradients, 1x1 / stride 2 / deconv)")
        v = groups.setdefault(gk, [0, 0.0, 0.0, 0.0])
        v[0] += 1; v[1] += flops; v[2] += flops * (1.5 if w4x3 else 0.25 if f4 else 16.0 / 36.0 if wino else 6.0 if x3p else 1.0); v[3] += e0.elapsed_time(e1) * 1e-3
    for flops, e0, e1, shp in wprof:
        v = groups["conv_wgrad (weight gradients)"]
        v[0] += 1; v[1] += flops; v[2] += flops; v[3] += e0.elapsed_time(e1) * 1e-3
    dom = max(groups, key=lambda kk: groups[kk][3])
    cnt, fl, exe, sec = groups[dom]
    tot_exe, tot_s = sum(v[2] for v in groups.values()), sum(v[3] for v in groups.values())
    # HBM-side bytes per launch of the dominant group from the committed PMC pass of this very command (tools/pmc_train_r04.sh: counters-only
    # FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE per the guide's gfx950 correction, averaged over the group's launches)
    traffic, traffic_note = None, "no PMC pass committed for this command (tools/pmc_train_r04.sh)"
    pmc_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"r04_pmc_train_{model_name}{'_amp' if amp else ''}.json")
    if os.path.exists(pmc_path):
        pm = json.load(open(pmc_path))
        gkey = "wino" if "wino" in dom else ("wgrad" if "wgrad" in dom else "igemm")
        if gkey in pm.get("per_group", {}):
            traffic = round(pm["per_group"][gkey]["bytes_per_launch"])
            traffic_note = f"profiles/{os.path.basename(pmc_path)}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, mean over {pm['per_group'][gkey]['launches']} launches of the group"
    pk = lambda kk: gpeak.get(kk, PEAK)
    busy = sum(v[2] / (pk(kk) * 1e12) for kk, v in groups.items())          # seconds at the peak of the pipe each group runs on
    res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(exe / sec / 1e12, 2), "peak": pk(dom), "unit": "TFLOP/s",
                       "frac": round(exe / sec / 1e12 / pk(dom), 4), "effective_tflops": round(fl / sec / 1e12, 2), "traffic": traffic, "traffic_note": traffic_note,
                       "launches_per_step": cnt / psteps, "avg_launch_us": round(sec / cnt * 1e6, 2),
                       "per_group": {kk: {"launches_per_step": v[0] / psteps, "ms_per_step": round(v[3] / psteps * 1e3, 3),
                                          "executed_tflops": round(v[2] / v[3] / 1e12, 2), "effective_tflops": round(v[1] / v[3] / 1e12, 2), "peak": pk(kk),
                                          "frac": round(v[2] / v[3] / 1e12 / pk(kk), 4)}
                                     for kk, v in groups.items() if v[0]},
                       "all_mfma_kernels": {"ms_per_step": round(tot_s / psteps * 1e3, 3), "executed_tflops": round(tot_exe / tot_s / 1e12, 2),
                                            "frac": round(busy / tot_s, 4), "frac_note": "time the groups would take at the peak of their own pipe / measured time"},
                       "note": "executed matrix-core FLOPs (Winograd launches: 16/36 of the direct count) over hipEvent pairs around every launch, "
                               f"{psteps} extra steps with the weight gradients on the main stream; the BatchNorm / pillar / attention kernels are HBM-bound "
                               "(tools/bn_bench.py) and not part of this object"}
    if a.cpu and dd_host is not None:
        from oracle import loss_oracle as lo
        from oracle import where2comm_oracle as orc
        sd2 = {kk: v.clone() for kk, v in sd.items()}
        for kk, v in sd2.items():
            if v.is_floating_point() and not kk.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        t0 = time.perf_counter()
        with orc.train_mode():
            o = orc.where2com_forward(dd_host, sd2, args, reference_schedule=True, topk=[H * W // 2])
        l = lo.pp_loss(o["psm"], o["rm"], o["obj"], tgt_host["targets"], tgt_host["pos_equal_one"], tgt_host["class_ids"], args["num_class"], 1.0, 2.0)
        l[0].backward()
        res["cpu_oracle_step_s"] = round(time.perf_counter() - t0, 2)
        res["cpu_threads"] = torch.get_num_threads()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu", action="store_true", help="also time one oracle step on the host cores")
    ap.add_argument("--small", action="store_true", help="128 x 64 canvas (smoke)")
    ap.add_argument("--model", choices=["where2com", "cobevt", "v2xvit", "when2com", "v2vnet"], default="where2com")
    ap.add_argument("--amp", action="store_true", help="autocast(bf16) + GradScaler around the step (tools/train.py --amp of the reference)")
    ap.add_argument("--modalities", default=None, help="e.g. cam,lidar or cam: the camera configuration (where2com only)")
    a = ap.parse_args()
    mods = a.modalities.split(",") if a.modalities else None
    print(json.dumps(run(a.agents, a.steps, a.warmup, a.small, a.cpu and a.model == "where2com" and not mods, model_name=a.model, amp=a.amp,
                         modalities=mods)))


if __name__ == "__main__":
    main()
