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
