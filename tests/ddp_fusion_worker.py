"""Worker of tests/test_gpu_train_cobevt.py::test_ddp_two_ranks (launched by torch.distributed.run, 2 ranks on ONE GPU over gloo): the
reference's data-parallel training wrap (tools/train.py:162, DistributedDataParallel(find_unused_parameters=True)) around
Airv2xCoBEVT / Airv2xV2XVit.  Each rank trains on its own frame; after backward every rank holds the mean of the single-process gradients."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.getcwd())
from tests.helpers import load_fixture  # noqa: E402


def main():
    which = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if which == "cobevt":
        from tests.test_gpu_train_cobevt import _case, _loss, _model
        names = ("train_cobevt_small_n3", "train_cobevt_small_n2")
    else:
        from tests.test_gpu_train_v2xvit import _case, _loss, _model
        names = ("train_v2xvit_small_n3", "train_v2xvit_small_n2")
    cases = [_case(load_fixture(n)) for n in names]
    args, sd = cases[0][1], cases[0][2]
    crit = _loss(args)

    def single(i):
        m = _model(args, sd)
        crit(m(cases[i][3]), cases[i][4]).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    g0, g1 = single(0), single(1)
    m = _model(args, sd)
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    crit(ddp(cases[rank][3]), cases[rank][4]).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in m.named_parameters():
        if k not in g0 and k not in g1:
            continue
        ref = (g0.get(k, torch.zeros_like(p)) + g1.get(k, torch.zeros_like(p))) / 2
        assert p.grad is not None, k
        worst = max(worst, float((p.grad - ref).abs().max()) / max(float(ref.abs().max()), 1e-30))
    assert worst < 1e-5, worst
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"DDP-2-OK {which} worst {worst:.2e}")


if __name__ == "__main__":
    main()
