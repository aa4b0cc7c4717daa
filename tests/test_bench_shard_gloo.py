"""bench.py's multi-rank leg (what the driver launches with --gpus N) driven over gloo on CPU, world 2 and 3:
the SAME code path as on the GPUs — partition_agents (4 agents over 3 ranks = [2,1,1]), ShardedPipeline with frames
in flight and the rotating ego stage, barrier + MAX-over-ranks timing, the secondary legs and the one JSON line —
with the compute swapped for the CPU oracle backend through bench.main(hooks=...).  The GPU engines implement the
same two-method backend (tests/test_gpu_sharded.py)."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc
from tests.test_sharded_gloo import RNG, OracleBackend


class OracleHooks:
    """bench.GpuShardHooks' interface on CPU: small grid, oracle voxelizer, oracle backend."""

    def __init__(self):
        self.log = []

    def inputs(self, n_agents, only):
        hy = synth.default_hypes(RNG)
        args = hy["model"]["args"]
        types = synth.sort_types(synth.agent_types_for(n_agents))[1]
        voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 300, RNG), RNG), RNG, [0.4, 0.4, 4.0])
                for i in range(n_agents)]
        dd = synth.build_data_dict([voxd[i] for i in only], [types[i] for i in only]) if len(only) else None
        self.log.append((n_agents, list(only)))
        self.full = (args, voxd, types)
        return hy, args, dd, types

    def backends(self, args, depth):
        self.sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
        return [OracleBackend(self.sd, args) for _ in range(depth)]


def _worker(rank, world, port, out_path, argv):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "LOCAL_WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    hooks = OracleHooks()
    res = bench.main(argv, hooks=hooks, device="cpu")
    if rank == 0:
        json.dump({"res": res, "log": hooks.log}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_multi_rank_default_is_the_agent_sharded_frame(tmp_path, world):
    out = str(tmp_path / "res.json")
    argv = ["--gpus", str(world), "--steps", "3", "--warmup", "1", "--inflight", "2", "--no-roofline"]
    mp.spawn(_worker, args=(world, _free_port(), out, argv), nprocs=world, join=True)
    got = json.load(open(out))
    res = got["res"]
    assert res["n_gpus"] == world and res["steps"] == 3 and res["warmup"] == 1
    assert res["scaling"] == "strong" and res["unit"] == "frames/s" and res["value"] > 0
    assert abs(res["value"] - 3 / (res["ms_per_step"] * 3e-3)) < 1e-2 * res["value"]
    assert "4-agent" in res["metric"] and "all_gather_into_tensor" in res["config"]["parallelism"]
    assert res["config"]["frames_in_flight"] == 2
    counts = {2: [2, 2], 3: [2, 1, 1]}[world]
    assert str(counts) in res["config"]["parallelism"]
    # round 6: the run describes its own exchange -- the process group's backend and rank count, the collective it took and every rank's
    # MEASURED message bytes (what `--dry-run` predicts), so that a SCALE record can be checked without the code
    par = res["config"]["parallelism"]
    assert "backend gloo" in par and f"{world} rank(s), collective" in par and "message bytes per rank and frame [" in par
    sizes = json.loads(par.split("message bytes per rank and frame ")[1].split("]")[0] + "]")
    assert len(sizes) == world and all(isinstance(v, int) and v > 0 for v in sizes) and len(set(sizes)) == 1      # padded to the largest rank
    assert "single_frame_latency" in res and res["single_frame_latency"]["ms_per_frame"] > 0
    # rank 0 was asked for its balanced slice of the 4-agent frame
    assert got["log"][0] == [4, list(range(counts[0]))]


def test_bench_more_ranks_than_four_agents_adds_the_four_agent_frame(tmp_path):
    """--agents 5 on 2 ranks ([3,2]) as the headline + the BASELINE 4-agent frame as a secondary figure."""
    out = str(tmp_path / "res.json")
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "0", "--inflight", "1", "--agents", "5", "--no-roofline"]
    mp.spawn(_worker, args=(2, _free_port(), out, argv), nprocs=2, join=True)
    res = json.load(open(out))["res"]
    assert "5-agent" in res["metric"] and "[3, 2]" in res["config"]["parallelism"]
    assert res["four_agent_frame"]["agents_per_rank"] == [2, 2]
