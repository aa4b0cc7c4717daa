"""GPU: the engine's agent-sharded stages.  Two "ranks" are emulated back to back on one GPU
(rank 0: agents 0-1 incl. the ego, rank 1: agents 2-3); their send buffers are concatenated the
way all_gather_into_tensor lays them out and the ego stage must reproduce the single-GPU forward
bit for bit (per-agent results do not depend on which other agents share the launch)."""
import pytest
import torch

from airv2x_perception_amd import synth
from tests.helpers import case_from_fixture, load_fixture

pytestmark = pytest.mark.gpu


def test_two_emulated_ranks_equal_single_gpu_forward():
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.sharded import EngineBackend, ShardedFrame, partition_agents
    fx = load_fixture("w2c_full_n4")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False   # bit-reproducible schedules only (stream-K splits K differently per agent count)
    ref = eng.forward(dd, sync_comm_rate=True)
    sends, stats, meta = [], None, None
    for r, mine in enumerate(partition_agents(4, 2)):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine])
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0))
        sends.append(send.clone())
        stats = st.clone() if stats is None else stats + st
    out = eng.shard_ego_stage(torch.cat(sends), stats, meta, world=2, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"] == int(fx["comm_rate"])
    assert abs(float(out["com"]) - float(ref["com"])) < 1e-7
    # world == 1 through the public wrapper
    one = ShardedFrame(EngineBackend(eng)).forward(dd, sync_comm_rate=True)
    assert torch.equal(one["psm"], ref["psm"]) and one["comm_rate"] == ref["comm_rate"]


def test_shard_stages_replayed_from_hipgraphs_equal_the_eager_stages():
    """engine.use_graph: the per-rank local stage (after the scatter) and the ego stage of the agent-sharded Where2Comm frame are captured
    once per layout and replayed (they are launch-bound from ~4 ranks on: bench.py --dry-run `launch_floor`).  Same bits as the eager
    stages, on every replay, and the graphs read the CURRENT frame's buffers (two different frames through the same graphs)."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture("w2c_full_n4")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    parts = partition_agents(4, 2)
    assert types[0] == types[1]
    frames = [voxd, [voxd[1], voxd[0], voxd[2], voxd[3]]]          # frame B: the two vehicles' clouds swapped (another ego)
    tys = [types, types]

    def run(frame, use_graph):
        eng.use_graph = use_graph
        recv = eng.buf("test_recv", (2 * eng.shard_local_stage(synth.build_data_dict([frames[frame][i] for i in parts[0]],
                                                                                      [tys[frame][i] for i in parts[0]]), has_ego=True)[0].numel(),))
        stats = None
        for r, mine in enumerate(parts):
            dd_local = synth.build_data_dict([frames[frame][i] for i in mine], [tys[frame][i] for i in mine])
            send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0))
            recv[r * send.numel():(r + 1) * send.numel()].copy_(send)
            stats = st.clone() if stats is None else stats + st
        out = eng.shard_ego_stage(recv, stats, meta, world=2, sync_comm_rate=True)
        return {k: out[k].clone() for k in ("psm", "rm", "obj")}, out["comm_rate"], float(out["com"])

    eager = [run(f, False) for f in (0, 1)]
    assert not torch.equal(eager[0][0]["psm"], eager[1][0]["psm"])
    for rep in range(2):
        for f in (0, 1):
            got = run(f, True)
            for k in ("psm", "rm", "obj"):
                assert torch.equal(got[0][k], eager[f][0][k]), (rep, f, k)
            assert got[1] == eager[f][1] and got[2] == eager[f][2]
    assert sum(1 for k in eng.graphs if k[0] in ("shard_local", "shard_ego")) >= 2
    eng.use_graph = False


@pytest.mark.parametrize("name,amp", [("cobevt_small_n2_c4", False), ("cobevt_small_n3", False), ("cobevt_small_n2_c4", True),
                                      ("cobevt_small_n3", True)])
def test_cobevt_emulated_ranks_equal_single_gpu_forward(name, amp):
    """CoBEVT: rank r runs the trunk (and, with compression, the NaiveCompressor encoder) of its agents; the
    gathered messages are decoded / regrouped and fused on the ego side.  n3 is emulated as 3 ranks x 1 agent.
    amp: the autocast frame -- the message is bf16 (half the bytes per link) and the sharded frame still has the single frame's bits."""
    import tests.test_cobevt as tc
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture(name)
    hy, args, sd, dd = tc._case(fx)
    types = [str(t) for t in fx["types"]]
    rng = [float(v) for v in fx["lidar_range"]]
    from oracle import voxelize_oracle as vox
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 hy["preprocess"]["args"]["voxel_size"]) for i in range(len(types))]
    model = Airv2xCoBEVT(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    eng.amp = amp
    ref = {k: v.clone() for k, v in eng.forward(dd).items()}
    world = len(types)
    sends, meta = [], None
    for r, mine in enumerate(partition_agents(len(types), world)):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], max_cav_num=args["max_cav_num"])
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0))
        sends.append(send.clone())
    if int(fx["compression"]) if "compression" in fx else 0:
        assert sends[0].numel() * 4 == meta["H"] * meta["W"] * 256   # the message is 4x smaller than the feature map
    assert sends[0].dtype == (torch.bfloat16 if amp else torch.float32)       # autocast: 2 bytes per element on the link
    out = eng.shard_ego_stage(torch.cat(sends), st, meta, world=world)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
        if not amp:
            tc.assert_close(out[k].cpu(), fx[k], 3e-4, 3e-4, k)
        else:   # the autocast drift bound of tests/test_amp.py
            assert float((out[k].cpu() - torch.from_numpy(fx[k])).abs().max()) <= 6e-2 * float(abs(fx[k]).max()), k
    # second level: every rank fuses only its residue-group columns of the map (no exchange between the window and the
    # grid halves), the head outputs are gathered.  64 columns = 4 groups: 2 + 2 (+ an all-padding third rank for n3)
    recv = torch.cat(sends)
    parts, ctx = [], None
    for r in range(world):
        part, ctx = eng.shard_ego_partial(recv, st, meta, world, r)
        parts.append(part.clone())
    out2 = eng.shard_ego_finish(torch.cat(parts), ctx, world)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out2[k], ref[k]), ("two-level", k)


@pytest.mark.parametrize("name,amp", [("v2xvit_small_n3", False), ("v2xvit_small_n3", True), ("v2xvit_full_n8", True)])
def test_v2xvit_emulated_ranks_equal_single_gpu_forward(name, amp):
    """amp: the autocast frame (bf16 activations, LayerNorm / attention fused into the Linears) shards to the same bits too"""
    import tests.test_v2xvit as tv
    from airv2x_perception_amd.opencood_iface import Airv2xV2XVit
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture(name)
    hy, args, sd, dd = tv._case(fx)
    types = [str(t) for t in fx["types"]]
    rng = [float(v) for v in fx["lidar_range"]]
    from oracle import voxelize_oracle as vox
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 hy["preprocess"]["args"]["voxel_size"]) for i in range(len(types))]
    model = Airv2xV2XVit(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    eng.amp = amp
    ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd, sync_comm_rate=True).items()}
    world = len(types)
    sends, stats, meta = [], None, None
    for r, mine in enumerate(partition_agents(len(types), world)):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], max_cav_num=args["max_cav_num"])
        for k in ("prior_encoding", "spatial_correction_matrix"):   # frame-level metadata of all agents
            dd_local[k] = dd[k]
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0))
        assert send.dtype == (torch.bfloat16 if amp else torch.float32)       # autocast: the 18.0 MB-per-agent message of SURVEY 8e
        sends.append(send.clone())
        stats = st.clone() if stats is None else stats + st
    out = eng.shard_ego_stage(torch.cat(sends), stats, meta, world=world, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"]
    # second level: 2 "ranks", each runs the encoder blocks on half of the columns; the only cross-rank quantity is the
    # split-attention mean of every block.  The emulation feeds each rank the global means recorded from a full run (what
    # the all-reduce would deliver) and checks separately that the ranks' local means average to them.
    recv = torch.cat(sends)
    eng.gap_record = []
    eng.shard_ego_stage(recv.clone(), stats, meta, world=world)
    global_gaps, eng.gap_record = eng.gap_record, None
    local_gaps = {}

    def exchange(gap, w, idx):
        local_gaps.setdefault(idx, []).append(gap.clone())
        gap.copy_(global_gaps[idx])
    eng.gap_exchange = exchange
    parts, ctx = [], None
    try:
        for r in range(2):
            part, ctx = eng.shard_ego_partial(recv.clone(), stats, meta, world, r, fusion_world=2, fusion_rank=r)
            parts.append(part.clone())
    finally:
        eng.gap_exchange = None
    out2 = eng.shard_ego_finish(torch.cat(parts), ctx, 2, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out2[k], ref[k]), ("two-level", k)
    assert out2["comm_rate"] == ref["comm_rate"]
    assert len(global_gaps) == 3 and sorted(local_gaps) == [0, 1, 2]
    for idx, g in enumerate(global_gaps):     # mean of the two strips' means = the global mean (equal strips)
        both = local_gaps[idx]
        assert len(both) == 2 and both[0].shape == g.shape
        torch.testing.assert_close((both[0] + both[1]) / 2, g, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("which", ["where2com", "cobevt", "v2xvit", "when2com"])
def test_frame_pipeline_results_equal_sequential_forward(which):
    """FramePipeline (frames in flight on separate HIP streams, shared packed weights, own workspaces) returns, for every
    frame, exactly what the sequential data-parallel forward returns."""
    from airv2x_perception_amd.opencood_iface.engine import FramePipeline
    if which == "where2com":
        from airv2x_perception_amd.opencood_iface import Airv2xWhere2com as M
        fx = load_fixture("w2c_small_n3")
        hy, args, sd, dd, _, _ = case_from_fixture(fx)
    elif which == "cobevt":
        import tests.test_cobevt as tc
        from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT as M
        fx = load_fixture("cobevt_small_n3")
        hy, args, sd, dd = tc._case(fx)
    elif which == "when2com":
        import tests.test_when2com as tw
        from airv2x_perception_amd.opencood_iface import Airv2xWhen2com as M
        fx = load_fixture("when2com_small_n3")
        hy, args, sd, dd = tw._case(fx)
    else:
        import tests.test_v2xvit as tv
        from airv2x_perception_amd.opencood_iface import Airv2xV2XVit as M
        fx = load_fixture("v2xvit_small_n3")
        hy, args, sd, dd = tv._case(fx)
    model = M(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    eng.throughput_mode = True          # the mode FramePipeline(depth > 1) puts the engine in (engine.wino4_rule): same mode, same bits
    ref = {k: v.clone() for k, v in eng.forward(dd).items() if torch.is_tensor(v) and v.dim() > 0}
    pipe = FramePipeline(eng, 3)
    outs = [pipe.submit(dd)[0] for _ in range(5)]
    pipe.drain()                        # the caller's stream waits for the frames' streams before touching the outputs
    torch.cuda.synchronize()
    outs = [{k: v for k, v in o.items() if torch.is_tensor(v) and v.dim() > 0} for o in outs[-3:]]   # one per engine
    for o in outs:
        for k in ("psm", "rm", "obj"):
            assert torch.equal(o[k], ref[k]), (which, k)


def test_when2com_emulated_ranks_equal_single_gpu_forward():
    import tests.test_when2com as tw
    from airv2x_perception_amd.opencood_iface import Airv2xWhen2com
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture("when2com_small_n3")
    hy, args, sd, dd = tw._case(fx)
    types = [str(t) for t in fx["types"]]
    rng = [float(v) for v in fx["lidar_range"]]
    from oracle import voxelize_oracle as vox
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 hy["preprocess"]["args"]["voxel_size"]) for i in range(len(types))]
    model = Airv2xWhen2com(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd, sync_comm_rate=True).items()}
    world = len(types)
    sends, stats, meta = [], None, None
    for r, mine in enumerate(partition_agents(len(types), world)):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], max_cav_num=args["max_cav_num"])
        dd_local["img_pairwise_t_matrix_collab"] = dd["img_pairwise_t_matrix_collab"]     # frame-level metadata
        dd_local["shard_rank"] = r
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0))
        sends.append(send.clone())
        stats = st.clone() if stats is None else stats + st
    out = eng.shard_ego_stage(torch.cat(sends), stats, meta, world=world, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"]


@pytest.mark.parametrize("n_agents,world", [(4, 3), (3, 2), (4, 8)])
def test_uneven_emulated_ranks_equal_single_gpu_forward(n_agents, world):
    """4 agents on 3 ranks ([2,1,1]), 3 on 2 ([2,1]), 4 on 8 (four idle ranks): padded messages, counts-driven fusion;
    bit-identical to the single-GPU forward."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture("w2c_full_n4")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    voxd, types = voxd[:n_agents], types[:n_agents]
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    ref = eng.forward(synth.build_data_dict(voxd, types), sync_comm_rate=True)
    parts = partition_agents(n_agents, world)
    counts = [len(p) for p in parts]
    n_pad = max(counts)
    sends, stats, meta = [], None, None
    for r, mine in enumerate(parts):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine]) if len(mine) else None
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0), n_pad=n_pad)
        sends.append(send.clone() if len(mine) else torch.full_like(send, float("nan")))   # padding must never be read
        stats = st.clone() if stats is None else stats + st
    meta = dict(meta, counts=counts)
    out = eng.shard_ego_stage(torch.cat(sends), stats, meta, world=world, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"]
    assert abs(float(out["com"]) - float(ref["com"])) < 1e-7


def test_rccl_group_of_one_rank_runs_the_all_gather_path():
    """A 1-rank "nccl" (= RCCL) process group on the box's one GPU: communicator set-up + all_gather_into_tensor +
    all_reduce as ShardedFrame issues them (world > 1 needs more GPUs; the driver's scaling run covers that)."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.sharded import EngineBackend, ShardedFrame
from tests.helpers import case_from_fixture, load_fixture
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hy, args, sd, dd, voxd, types = case_from_fixture(load_fixture("w2c_small_n3"))
m = Airv2xWhere2com(args); m.load_state_dict(sd); m = m.to("cuda").eval()
eng = m.engine(); eng.stream_k = False
ref = eng.forward(dd, sync_comm_rate=True)
out = ShardedFrame(EngineBackend(eng), collectives_when_single=True).forward(dd, counts=[len(types)], sync_comm_rate=True)
torch.cuda.synchronize()
assert all(torch.equal(out[k], ref[k]) for k in ("psm", "rm", "obj")) and out["comm_rate"] == ref["comm_rate"]
# the opt-in route of the rotating ego stage: dist.gather + dist.reduce to the fusion rank on the RCCL communicator
os.environ["AV2X_SHARD_GATHER"] = "1"
out2 = ShardedFrame(EngineBackend(eng), collectives_when_single=True).forward(dd, counts=[len(types)], fusion_rank=0, sync_comm_rate=True)
torch.cuda.synchronize()
assert all(torch.equal(out2[k], ref[k]) for k in ("psm", "rm", "obj")) and out2["comm_rate"] == ref["comm_rate"]
dist.destroy_process_group()
print("RCCL-1-OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-1-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("extra", [[], ["--agents", "3"]])
def test_bench_two_ranks_on_one_gpu_reports_the_sharded_frame(extra):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run), both ranks on this box's one GPU
    over gloo (RCCL refuses two ranks per device): the JSON line is the agent-sharded frame."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AV2X_ONE_DEVICE="1", AV2X_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-roofline"] + extra
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    assert "all_gather_into_tensor" in res["config"]["parallelism"]
    assert ("[2, 1]" if extra else "[2, 2]") in res["config"]["parallelism"]
    assert res["replica"]["frames_per_s"] > 0 and res["single_frame_latency"]["frames_per_s"] > 0


@pytest.mark.parametrize("which,ratio,amp,uneven", [("v2xvit", 2, False, False), ("v2xvit", 4, True, False), ("v2xvit", 4, False, True),
                                                    ("when2com", 4, False, False), ("when2com", 2, False, True)])
def test_sharded_frame_with_a_naive_compressor_equals_the_single_gpu_forward(which, ratio, amp, uneven):
    """V2X-ViT / When2com with ``modality_fusion.compression > 0`` (ratio = args["compression"]; airv2x_v2xvit.py:42-44,122-123,
    airv2x_when2com.py:50-52,122-123) in the agent-sharded frame.  V2X-ViT: the message is the compressor's ENCODER output -- 256 / ratio
    channels per pixel, ratio x fewer bytes through the all-gather -- and the receiver runs the decoder; When2com: its per-agent work (warp,
    policy network, keys) stays on the sender, so encoder + decoder run there and the message is unchanged.  Same bits as the unsharded frame."""
    from airv2x_perception_amd import opencood_iface as oi
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    from oracle import voxelize_oracle as vox
    rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
    types = ["vehicle", "vehicle", "rsu"] if uneven else ["vehicle", "rsu"]
    if which == "v2xvit":
        hy, spec_fn, cls = synth.default_hypes_v2xvit(rng, (2, 1, 1)), synth.v2xvit_param_spec, oi.Airv2xV2XVit
    else:
        hy, spec_fn, cls = synth.default_hypes_when2com(rng), synth.when2com_param_spec, oi.Airv2xWhen2com
    args = hy["model"]["args"]
    args["modality_fusion"]["compression"] = args["compression"] = ratio
    sd = synth.synthetic_state_dict(spec_fn(args), seed=40 + ratio)
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 900, rng), pp["cav_lidar_range"]), pp["cav_lidar_range"],
                                 pp["args"]["voxel_size"]) for i in range(len(types))]
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    if which == "v2xvit":
        scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
        for i in range(1, len(types)):
            scm[0, i] = torch.from_numpy(synth.se2_correction(2.0 * i, 0.6 * i, -0.3 * i))
        dd["spatial_correction_matrix"] = scm
    else:
        dd["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(len(types), args["max_cav_num"])
    model = cls(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    eng.amp = amp
    ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd, sync_comm_rate=True).items()}
    world = 2
    parts = partition_agents(len(types), world)
    counts = [len(p) for p in parts]
    sends, stats, meta = [], None, None
    for r, mine in enumerate(parts):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], max_cav_num=args["max_cav_num"])
        for k in ("prior_encoding", "spatial_correction_matrix", "img_pairwise_t_matrix_collab"):
            if k in dd:
                dd_local[k] = dd[k]
        dd_local["shard_rank"] = r
        dd_local["shard_agent_offset"] = sum(counts[:r])
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0), n_pad=max(counts))
        if which == "v2xvit":       # the payload: n_pad x H x W x 256 / ratio elements of 4 (autocast: 2) bytes
            assert send.numel() == max(counts) * meta["H"] * meta["W"] * (256 // ratio)
            assert send.dtype == (torch.bfloat16 if amp else torch.float32)
        sends.append(send.clone())
        stats = st.clone() if stats is None else stats + st
    out = eng.shard_ego_stage(torch.cat(sends), stats, dict(meta, counts=counts, n_pad=max(counts)), world=world, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"]
