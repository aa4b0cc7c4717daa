"""Training branch of Communication.forward (where2comm_fuse.py:104-121): every agent transmits its K most
confident cells, K = int(H * W * random.uniform(0, 1)) per sample.  Golden = the reference's own module in .train() mode
with Python's `random` seeded (tools/gen_golden.py `comm_train`)."""
import random

import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import where2comm_oracle as orc
from tests.helpers import load_fixture

CFG = {"round": 1, "threshold": 0.01, "gaussian_smooth": {"k_size": 5, "c_sigma": 1.0}}


def _inputs(fx):
    lens = [int(v) for v in fx["lens"]]
    H, W = int(fx["H"]), int(fx["W"])
    psm = torch.from_numpy(synth.seeded_uniform(int(fx["seed_psm"]), (sum(lens), 14, H, W), -6.0, 2.0))
    sd = {"fusion_net.naive_communication.gaussian_filter.weight": torch.from_numpy(fx["gauss_w"]),
          "fusion_net.naive_communication.gaussian_filter.bias": torch.from_numpy(fx["gauss_b"])}
    return lens, H, W, psm, sd


def test_oracle_training_branch_matches_reference_golden():
    fx = load_fixture("comm_train")
    lens, H, W, psm, sd = _inputs(fx)
    for trial, seed in enumerate((1, 2, 3)):
        random.seed(seed)
        ks = [int(H * W * random.uniform(0, 1)) for _ in lens]          # the reference's draw (:106)
        assert ks == [int(v) for v in fx[f"k_{trial}"]]
        m, rate, _ = orc.communication(list(torch.split(psm, lens)), sd, CFG, topk=ks)
        assert np.array_equal(np.packbits(m.numpy().astype(np.uint8).reshape(-1)), fx[f"mask_{trial}"])
        assert float(rate) == float(fx[f"rate_{trial}"])


@pytest.mark.gpu
def test_gpu_topk_mask_matches_reference_golden():
    from airv2x_perception_amd.opencood_iface.submodules import _Runner, _nhwc
    fx = load_fixture("comm_train")
    lens, H, W, psm, sd = _inputs(fx)
    r = _Runner(torch.device("cuda"), fcfg={"communication": CFG, "fully": False, "multi_scale": True})
    r._load_fusion(sd, r._up)
    r.A, r.C = 14, 1
    x = _nhwc(psm.cuda())
    for trial in range(3):
        ks = [int(v) for v in fx[f"k_{trial}"]]
        mask, count, smooth, rl = r.comm_mask(x, sum(lens), H, W, lens, topk=ks)
        got = mask.cpu().numpy().astype(np.uint8).reshape(-1)
        assert np.array_equal(np.packbits(got), fx[f"mask_{trial}"])
        assert count.cpu().tolist() == [k * n for k, n in zip(ks, lens)]      # cells set before the ego override
        rate = r.comm_rate(count, rl, len(lens), H * W)
        assert abs(float(rate) - float(fx[f"rate_{trial}"])) < 1e-6
    # edge values of K
    for ks in ([0, H * W], [1, H * W - 1]):
        mask, count, _, _ = r.comm_mask(x, sum(lens), H, W, lens, topk=ks)
        m, _, _ = orc.communication(list(torch.split(psm, lens)), sd, CFG, topk=ks)
        assert np.array_equal(mask.cpu().numpy().reshape(-1), m.numpy().reshape(-1)) and count.cpu().tolist() == [k * n for k, n in zip(ks, lens)]
