"""The 8-rank layouts BASELINE.json's north_star names, executed for real (8 processes, gloo, CPU oracle backends) before any
8-GPU node sees them:

  * Where2Comm: 8 agents / 8 ranks (one per rank), 4 agents / 8 ranks (four idle ranks send only padding), 5 agents / 8 ranks,
    two frames in flight with the ego stage of frame t on rank t % 8, all-gather and the opt-in gather-to-the-fusion-rank;
  * CoBEVT two-level sharding on the DEFAULT grid width (352 feature columns = 22 residue groups over 8 ranks: 3 groups each, the last
    rank 1 real + 2 padded) and on a narrow map where most ranks hold nothing but padding, with uneven agent counts;
  * V2X-ViT two-level column strips (352 % (4 * 8) == 0: 44-column strips, one all-reduce per block), and 4 agents / 8 ranks;
  * bench.py's own multi-rank leg at --gpus 8 (what the driver launches), 8 / 4 / 5 agents.

Every case is compared with the single-process forward of the same frame.  The backends are the oracle classes of
tests/test_sharded_gloo.py (NaN padding: a fusion that read a padding slot would fail); the protocol under test is
opencood_iface/sharded.py + bench.shard_leg, the code the GPUs run.
"""
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.sharded import ShardedFrame, ShardedPipeline, partition_agents
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc
from tests.test_sharded_gloo import RNG, CoBEVTOracleBackend, OracleBackend, V2XViTOracleBackend, _free_port

WORLD = 8
WIDE = [-140.8, -3.2, -3.0, 140.8, 3.2, 1.0]      # 704 x 16 pillars: the default grid's width (352 feature columns), 8 feature rows


def _types(n):
    return synth.sort_types(synth.agent_types_for(n))[1]


def _voxels(n, rng, pts=400):
    return [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, pts, rng), rng), rng, [0.4, 0.4, 4.0]) for i in range(n)]


def _init(rank, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)


def _local(voxd, types, mine, **kw):
    return synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], **kw)


# ------------------------------------------------------------------------------------------------------- Where2Comm
def _w2c_worker(rank, port, path, n_agents, gather, depth):
    if gather is None:
        os.environ.pop("AV2X_SHARD_GATHER", None)        # the default: gather to the fusion rank from 4 ranks on
    else:
        os.environ["AV2X_SHARD_GATHER"] = "1" if gather else "0"
    _init(rank, port)
    hy = synth.default_hypes(RNG)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
    types, voxd = _types(n_agents), _voxels(n_agents, RNG)
    parts = partition_agents(n_agents, WORLD)
    counts = [len(p) for p in parts]
    dd_local = _local(voxd, types, parts[rank]) if len(parts[rank]) else None
    pipe = ShardedPipeline([OracleBackend(sd, args) for _ in range(depth)], rotate=True)
    assert all(f.gather_to_fusion_rank == (True if gather is None else gather) for f in pipe.frames)
    outs = []
    with torch.no_grad():
        for t in range(WORLD):
            outs.append(pipe.submit(dd_local, counts=counts)[0])
    pipe.drain()
    assert [o is not None for o in outs] == [t == rank for t in range(WORLD)]      # frame t finishes on rank t % 8 only
    torch.save({k: outs[rank][k] for k in ("psm", "rm", "obj", "com", "comm_rate")}, f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_agents,gather,depth", [(8, False, 2), (4, None, 2), (5, True, 2)])
def test_where2comm_world8(tmp_path, n_agents, gather, depth):
    path = str(tmp_path / "o")
    mp.spawn(_w2c_worker, args=(_free_port(), path, n_agents, gather, depth), nprocs=WORLD, join=True)
    hy = synth.default_hypes(RNG)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
    dd = synth.build_data_dict(_voxels(n_agents, RNG), _types(n_agents))
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args)
    for r in range(WORLD):
        got = torch.load(f"{path}.{r}")
        for k in ("psm", "rm", "obj"):
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), (r, k)
        assert got["comm_rate"] == ref["comm_rate"] and abs(float(got["com"]) - float(ref["com"])) < 1e-6


# ------------------------------------------------------------------------------------------------- CoBEVT / V2X-ViT
def _cobevt_case(rng):
    hy = synth.default_hypes_cobevt(rng, max_cav=(4, 2, 2))
    args = hy["model"]["args"]
    return args, synth.synthetic_state_dict(synth.cobevt_param_spec(args), seed=3)


def _v2xvit_case(rng, n_agents, compression=0):
    hy = synth.default_hypes_v2xvit(rng)
    args = hy["model"]["args"]
    if compression:     # NaiveCompressor: the sharded message is its encoder output (256 / ratio channels), decoded on the receiver
        args["modality_fusion"]["compression"] = args["compression"] = int(compression)
    sd = synth.synthetic_state_dict(synth.v2xvit_param_spec(args), seed=5)
    types, voxd = _types(n_agents), _voxels(n_agents, rng, 300)
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
    for i in range(1, n_agents):
        scm[0, i] = torch.from_numpy(synth.se2_correction(1.5 * i, 0.4 * i, -0.2 * i))
    dd["spatial_correction_matrix"] = scm
    return args, sd, types, voxd, dd


def _two_level_worker(rank, port, path, model, wide, n_agents):
    _init(rank, port)
    rng = WIDE if wide else RNG
    parts = partition_agents(n_agents, WORLD)
    counts = [len(p) for p in parts]
    if model == "cobevt":
        args, sd = _cobevt_case(rng)
        types, voxd = _types(n_agents), _voxels(n_agents, rng, 300)
        dd_local = _local(voxd, types, parts[rank]) if len(parts[rank]) else None
        backend = CoBEVTOracleBackend(sd, args, two_level=True)
    else:
        args, sd, types, voxd, dd = _v2xvit_case(rng, n_agents)
        dd_local = _local(voxd, types, parts[rank], max_cav_num=args["max_cav_num"])      # idle ranks: an empty frame + the frame metadata
        for k in ("prior_encoding", "spatial_correction_matrix"):
            dd_local[k] = dd[k]
        backend = V2XViTOracleBackend(sd, args, two_level=True)
    with torch.no_grad():
        out = ShardedFrame(backend).forward(dd_local, counts=None if len(set(counts)) == 1 else counts)
    torch.save({k: out[k] for k in ("psm", "rm", "obj")}, f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model,wide,n_agents", [("cobevt", True, 8), ("cobevt", False, 5), ("v2xvit", True, 8), ("v2xvit", False, 4)])
def test_two_level_fusion_sharding_world8(tmp_path, model, wide, n_agents):
    """wide: the default grid's 352 feature columns (CoBEVT: 22 residue groups -> 3 per rank, rank 7 holds 1 real + 2 padded; V2X-ViT:
    44-column strips).  narrow (32 columns): CoBEVT has 2 groups for 8 ranks (six ranks fuse padding only), V2X-ViT 4-column strips."""
    path = str(tmp_path / "o")
    mp.spawn(_two_level_worker, args=(_free_port(), path, model, wide, n_agents), nprocs=WORLD, join=True)
    rng = WIDE if wide else RNG
    if model == "cobevt":
        from oracle import cobevt_oracle as cob
        from airv2x_perception_amd.opencood_iface.sharded import fusion_column_shards
        args, sd = _cobevt_case(rng)
        dd = synth.build_data_dict(_voxels(n_agents, rng, 300), _types(n_agents))
        with torch.no_grad():
            ref = cob.cobevt_forward(dd, sd, args)
        W = ref["psm"].shape[-1]
        valid = [v for _, v in fusion_column_shards(W, 4, WORLD)]
        assert valid == ([12] * 7 + [4] if wide else [4, 4] + [0] * 6)
    else:
        from oracle import v2xvit_oracle as vit
        args, sd, types, voxd, dd = _v2xvit_case(rng, n_agents)
        with torch.no_grad():
            ref = vit.v2xvit_forward(dd, sd, args)
        assert ref["psm"].shape[-1] % (4 * WORLD) == 0
    for r in range(WORLD):
        got = torch.load(f"{path}.{r}")
        for k in ("psm", "rm", "obj"):
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), (model, r, k, float((got[k] - ref[k]).abs().max()))


def _v2xvit_compressed_worker(rank, port, path, n_agents, ratio):
    _init(rank, port)
    parts = partition_agents(n_agents, WORLD)
    counts = [len(p) for p in parts]
    args, sd, types, voxd, dd = _v2xvit_case(RNG, n_agents, ratio)
    dd_local = _local(voxd, types, parts[rank], max_cav_num=args["max_cav_num"])
    for k in ("prior_encoding", "spatial_correction_matrix"):
        dd_local[k] = dd[k]
    frame = ShardedFrame(V2XViTOracleBackend(sd, args, two_level=False))
    with torch.no_grad():
        out = frame.forward(dd_local, counts=None if len(set(counts)) == 1 else counts)
    torch.save({**{k: out[k] for k in ("psm", "rm", "obj")}, "message_bytes": frame.last_exchange["message_bytes"],
                "collective": frame.last_exchange.get("collective")}, f"{path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_agents,ratio", [(8, 4), (5, 2)])
def test_v2xvit_compressed_message_world8(tmp_path, n_agents, ratio):
    """V2X-ViT with a NaiveCompressor over 8 ranks (one agent per rank; 5 agents: three idle ranks send padding): every rank's message is
    the encoder output of ITS agents -- 256 / ratio channels per pixel, ratio x fewer bytes through the exchange than the uncompressed
    frame -- and every rank's result equals the single-process oracle of the compressed model."""
    from oracle import v2xvit_oracle as vit
    path = str(tmp_path / "o")
    mp.spawn(_v2xvit_compressed_worker, args=(_free_port(), path, n_agents, ratio), nprocs=WORLD, join=True)
    args, sd, types, voxd, dd = _v2xvit_case(RNG, n_agents, ratio)
    with torch.no_grad():
        ref = vit.v2xvit_forward(dd, sd, args)
    H, W = ref["psm"].shape[-2:]
    for r in range(WORLD):
        got = torch.load(f"{path}.{r}")
        assert got["message_bytes"] == 1 * (256 // ratio) * H * W * 4, (r, got["message_bytes"])      # n_pad = 1 agent slot per rank
        for k in ("psm", "rm", "obj"):
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), (r, k, float((got[k] - ref[k]).abs().max()))


# ------------------------------------------------------------------------------------------------- bench.py --gpus 8
def _bench_worker(rank, port, out_path, argv):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(WORLD),
                       "LOCAL_RANK": str(rank), "LOCAL_WORLD_SIZE": str(WORLD)})
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    import bench
    from tests.test_bench_shard_gloo import OracleHooks
    hooks = OracleHooks()
    res = bench.main(argv, hooks=hooks, device="cpu")
    if rank == 0:
        json.dump({"res": res, "log": hooks.log}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("agents,counts", [(0, [1] * 8), (5, [1] * 5 + [0] * 3)])
def test_bench_gpus8_line(tmp_path, agents, counts):
    """`bench.py --gpus 8` as the driver launches it: the default is ONE 8-agent frame over the 8 ranks (one agent per GPU); the BASELINE
    4-agent frame (ranks 4-7 idle) is reported as a secondary figure; --agents 5 leaves three ranks idle in the headline itself."""
    out = str(tmp_path / "res.json")
    argv = ["--gpus", "8", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-roofline"] + (["--agents", str(agents)] if agents else [])
    mp.spawn(_bench_worker, args=(_free_port(), out, argv), nprocs=WORLD, join=True)
    got = json.load(open(out))
    res = got["res"]
    n = agents or 8
    assert res["n_gpus"] == 8 and res["scaling"] == "strong" and res["value"] > 0 and f"{n}-agent" in res["metric"]
    assert str(counts) in res["config"]["parallelism"] and res["config"]["frames_in_flight"] == 2
    assert abs(res["value"] - res["steps"] / (res["ms_per_step"] * res["steps"] * 1e-3)) < 1e-2 * res["value"]
    assert res["four_agent_frame"]["agents_per_rank"] == [1, 1, 1, 1, 0, 0, 0, 0]
    assert res["single_frame_latency"]["ms_per_frame"] > 0
    assert got["log"][0] == [n, [0]]              # rank 0 holds the ego


# ------------------------------------------------------------------------------------------------- bench.py --dry-run
@pytest.mark.parametrize("model,agents", [("where2com", 0), ("where2com", 4), ("where2com", 5), ("cobevt", 8), ("v2xvit", 8)])
def test_bench_dry_run_describes_the_8_rank_layout(model, agents, capsys):
    import bench
    argv = ["--gpus", "8", "--dry-run", "--model", model] + (["--agents", str(agents)] if agents else [])
    r = bench.main(argv)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line == json.loads(json.dumps(r)) and r["dry_run"] and r["world"] == 8
    n = agents or 8
    assert sum(r["agents_per_rank"]) == n and r["n_pad"] == 1 and r["idle_ranks"] == list(range(n, 8))
    per = {"where2com": 15769600, "cobevt": 36044800, "v2xvit": 36044800}[model]
    assert r["bytes_per_agent"] == per and r["all_gather"]["bytes_per_link_per_frame"] == per
    assert r["all_gather"]["recv_bytes_per_rank"] == 8 * per and r["all_gather"]["padding_bytes_per_rank"] == (8 - n) * per
    if model == "where2com":
        assert r["autocast_message"] is None
    else:   # the 18.0 MB bf16 message SURVEY 8e quotes for the autocast frame
        assert r["autocast_message"]["bytes_per_agent"] == 18022400 and r["autocast_message"]["bytes_per_link_per_frame"] == 18022400
    if model == "cobevt":
        assert r["second_level"]["groups"] == 22 and r["second_level"]["valid_columns_per_strip"] == [12] * 7 + [4]
        assert r["second_level"]["padded_groups"] == 2
    if model == "v2xvit":
        assert r["second_level"]["splits"] and r["second_level"]["strip_width"] == 44
    r3 = bench.main(["--gpus", "3", "--dry-run", "--model", "v2xvit"])
    capsys.readouterr()
    assert not r3["second_level"]["splits"] and r3["agents_per_rank"] == [2, 1, 1]
