"""CPU: the train-mode oracle (oracle/where2comm_oracle.py under ``train_mode()`` + loss_oracle.pp_loss + torch autograd)
reproduces one training step of the REFERENCE model (tests/golden/train_small_*.npz, tools/gen_golden.py:train_golden):
head maps, losses, the gradient of every parameter and every BatchNorm buffer after the step."""
import numpy as np
import pytest

from oracle import loss_oracle as lo
from oracle import where2comm_oracle as orc
from tests.helpers import load_fixture, train_case_from_fixture


def oracle_step(args, sd, dd, tgt, K, loss_args=(7, 1.0, 2.0)):
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    with orc.train_mode():
        o = orc.where2com_forward(dd, sd, args, reference_schedule=True, topk=K)
    losses = lo.pp_loss(o["psm"], o["rm"], o["obj"], tgt["targets"], tgt["pos_equal_one"], tgt["class_ids"], *loss_args)
    losses[0].backward()
    return o, losses, sd


@pytest.mark.parametrize("name", ["train_small_n3", "train_small_n2", "train_small_single_c2", "train_small_single", "train_small_multi_c4"])
def test_train_oracle_reproduces_the_reference_step(name):
    fx = load_fixture(name)
    hy, args, sd, dd, tgt = train_case_from_fixture(fx)
    o, losses, sd2 = oracle_step(args, sd, dd, tgt, [int(k) for k in fx["K"]])
    hs = int(fx["head_stride"])
    for k in ("psm", "rm", "obj"):
        assert np.abs(o[k].detach()[..., ::hs, ::hs].numpy() - fx[k]).max() < 1e-5, k
    assert abs(float(losses[0].detach()) - fx["losses"][0]) < 1e-4 * abs(fx["losses"][0])
    assert abs(float(losses[1].detach()) - fx["losses"][1]) < 1e-4 * abs(fx["losses"][1])
    assert abs(float(losses[2].detach()) - fx["losses"][2]) < 1e-4 * abs(fx["losses"][2])
    for k in [str(k) for k in fx["grad_keys"]]:
        g = sd2[k].grad.reshape(-1)
        stride = max(1, g.numel() // 4096)
        gmax = fx["gsum:" + k][2]
        assert np.abs(g[::stride].numpy() - fx["g:" + k]).max() <= 1e-4 * gmax + 1e-9, k
        assert abs(g.double().abs().sum().item() - fx["gsum:" + k][1]) <= 1e-4 * fx["gsum:" + k][1] + 1e-9, k
    for k in fx.files:
        if k.startswith("b:"):
            ref = fx[k]
            got = sd2[k[2:]].detach().numpy()
            assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
