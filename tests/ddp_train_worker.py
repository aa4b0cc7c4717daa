"""Worker of tests/test_gpu_train.py::test_ddp_two_ranks_average_the_gradients (launched by torch.distributed.run, 2 ranks on
ONE GPU over gloo -- RCCL refuses two ranks per device): the reference's data-parallel training wrap (tools/train.py:162,
DistributedDataParallel(find_unused_parameters=True)) around the MI355X module.  Each rank trains on its own frame; after
backward every rank must hold the mean of the two single-process gradients."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.getcwd())
from airv2x_perception_amd.opencood_iface.airv2x_where2com import Airv2xWhere2com  # noqa: E402
from airv2x_perception_amd.opencood_iface.loss import PointPillarLossMultiClass  # noqa: E402
from airv2x_perception_amd.opencood_iface.train_where2com import forward_train  # noqa: E402
from tests.helpers import load_fixture, train_case_from_fixture  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cases = [train_case_from_fixture(load_fixture(n)) for n in ("train_small_n3", "train_small_n2")]
    args, sd = cases[0][1], cases[0][2]
    crit = PointPillarLossMultiClass({"cls_weight": 1.0, "reg": 2.0, "num_class": args["num_class"]})
    K = [[500], [900]]

    def single(i):
        m = Airv2xWhere2com(args)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = forward_train(m, cases[i][3], topk=K[i])
        crit(out, {k: v.cuda() for k, v in cases[i][4].items()}).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    g0, g1 = single(0), single(1)
    m = Airv2xWhere2com(args)
    m.load_state_dict(sd)
    m = m.cuda().train()
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    # DDP calls module.forward(data_dict), which draws K = int(H * W * random.uniform(0, 1)) as the reference does: pin the draw
    import random
    H, W = cases[rank][4]["pos_equal_one"].shape[1:3]
    real_uniform = random.uniform
    random.uniform = lambda lo, hi: (K[rank][0] + 0.5) / float(H * W)
    try:
        out = ddp(cases[rank][3])
    finally:
        random.uniform = real_uniform
    crit(out, {k: v.cuda() for k, v in cases[rank][4].items()}).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in m.named_parameters():
        if k not in g0 and k not in g1:
            continue
        a = g0.get(k, torch.zeros_like(p))           # a parameter one rank's frame does not touch contributes zero
        b = g1.get(k, torch.zeros_like(p))
        ref = (a + b) / 2
        assert p.grad is not None, k
        err = float((p.grad - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        worst = max(worst, err)
    assert worst < 1e-5, worst
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"DDP-2-OK worst {worst:.2e}")


if __name__ == "__main__":
    main()
