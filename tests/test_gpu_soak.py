"""Soak of the pipelined frame path (what bench.py's headline times): frames of two DIFFERENT clouds alternating in runs of seven through a
3-deep FramePipeline, every frame compared bit for bit (heads + comm_rate) with the one-frame-at-a-time result of the same engine mode.
Stale occupancy bytes of the sparse first convolution, non-zero counters, workspace races between the in-flight engines or a kernel that
mis-behaves next to another frame's kernels (the packed-fp32 / bf16-MFMA hazard of build.py) would all show here.
The long form (3 000 frames, 4 and 8 agents) is tools/soak_pipeline.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("agents,frames", [(4, 420), (8, 210)])
def test_pipelined_frames_alternating_clouds_bit_equal(agents, frames):
    from types import SimpleNamespace

    import bench
    from airv2x_perception_amd.opencood_iface.engine import FramePipeline
    dev = torch.device("cuda", 0)
    a = SimpleNamespace(model="where2com", amp=False, gemm="x3", agents=agents, points=8192, mods=("lidar",))
    _, args, dd, _, _ = bench.build_inputs(agents, 8192, dev, only=None, model="where2com", modalities=("lidar",))
    _, _, dd2, _, _ = bench.build_inputs(agents, 6000, dev, only=None, model="where2com", modalities=("lidar",))
    model, eng, _ = bench.make_model(a, args, dev)
    eng.throughput_mode = True
    refs = []
    for d in (dd, dd2):
        o = model(d)
        torch.cuda.synchronize()
        refs.append({**{k: o[k].clone() for k in ("psm", "rm", "obj")}, "comm_rate": int(o["comm_rate"])})
    assert not torch.equal(refs[0]["psm"], refs[1]["psm"])          # the two clouds really differ
    pipe = FramePipeline(eng, 3)
    pend, bad = [], []
    for f in range(frames):
        w = (f // 7) % 2
        pend.append((f, w) + tuple(pipe.submit(dd if w == 0 else dd2)[:2]))
        if len(pend) == 3:
            f0, w0, o, ev = pend.pop(0)
            ev.synchronize()
            if not (all(torch.equal(o[k], refs[w0][k]) for k in ("psm", "rm", "obj")) and int(o["comm_rate"]) == refs[w0]["comm_rate"]):
                bad.append(f0)
    pipe.drain()
    assert not bad, f"{len(bad)} of {frames} pipelined frames differ from the single-stream frame (first: {bad[:5]})"
