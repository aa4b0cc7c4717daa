"""When2com: CPU oracle vs reference golden; GPU engine vs golden + oracle.

Tolerances (fp32): oracle vs reference 1e-5; GPU vs reference 3e-4 relative + 3e-4 absolute on maps whose magnitude
is O(1..10) (different summation order in the convolutions / the 11k-input MLP); the attention coefficients 1e-4."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox
from oracle import when2com_oracle as w2
from tests.helpers import assert_close, load_fixture, sample

NAMES = ["when2com_small_n3", "when2com_small_n2"]


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_when2com(rng, mode=str(fx["mode"]))
    args = hy["model"]["args"]
    spec = synth.when2com_param_spec(args)
    assert [k for k, _, _ in spec] == [str(k) for k in fx["spec_keys"]]
    sd = synth.synthetic_state_dict(spec, seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 pp["args"]["voxel_size"]) for i in range(len(types))]
    for i, v in enumerate(voxd):
        assert np.array_equal(v[1], fx[f"vox_coords_{i}"])
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    dd["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(len(types), args["max_cav_num"])
    return hy, args, sd, dd


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(name):
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    tr = {}
    with torch.no_grad():
        out = w2.when2com_forward(dd, sd, args, trace=tr)
    for k in ("psm", "rm", "obj"):
        assert_close(sample(out[k], int(fx["head_stride"])), fx[k], 1e-5, 1e-5, k)
    assert_close(sample(tr["fused"], int(fx["big_stride"])), fx["fused"], 1e-5, 1e-5, "fused")
    assert_close(tr["coef0"].numpy(), fx["coef"], 1e-5, 1e-6, "coef")
    assert out["comm_rate"] == float(fx["comm_rate"]) and out["mask"] == 0
    assert abs(float(tr["coef0"].sum()) - 1.0) < 1e-5


def test_normalized_pairwise_host_matches_oracle():
    from airv2x_perception_amd.opencood_iface.when2com_engine import normalized_pairwise
    t = synth.when2com_pairwise(3, 5)
    a = normalized_pairwise(t.numpy(), 32, 64, 0.4, 4)
    b = w2.normalized_pairwise(t, 32, 64, 0.4, 4).numpy()
    assert a.shape == (1, 5, 5, 2, 3) and a.dtype == np.float32
    np.testing.assert_array_equal(a, b)


def test_activated_mode_is_refused():
    hy = synth.default_hypes_when2com([-25.6, -12.8, -3, 25.6, 12.8, 1], mode="activated")
    with pytest.raises(NotImplementedError):
        w2.when2com_fuse(torch.zeros(1, 256, 32, 64), torch.tensor([1]), torch.eye(4).view(1, 1, 1, 4, 4),
                         synth.synthetic_state_dict(synth.when2com_fusion_spec(hy["model"]["args"]["when2com_fusion"], "fusion_net.")),
                         hy["model"]["args"]["when2com_fusion"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES + ["when2com_full_n2"])
def test_gpu_forward_matches_golden(name):
    from airv2x_perception_amd.opencood_iface import Airv2xWhen2com, create_model
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    model = create_model(hy)
    assert isinstance(model, Airv2xWhen2com)
    assert list(model.state_dict().keys()) == [str(k) for k in fx["spec_keys"]]
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    tr = {}
    out = model.engine().forward(dd, trace=tr, sync_comm_rate=True)
    torch.cuda.synchronize()
    assert_close(tr["coef0"].cpu().numpy(), fx["coef"], 1e-4, 1e-5, "coef")
    assert_close(sample(tr["fused"], int(fx["big_stride"])), fx["fused"], 3e-4, 3e-4, "fused")
    for k in ("psm", "rm", "obj"):
        assert_close(sample(out[k], int(fx["head_stride"])), fx[k], 3e-4, 3e-4, k)
        tot, ref = float(out[k].double().sum()), float(fx[k + "_sum"])
        assert abs(tot - ref) <= 1e-5 * max(1.0, float(out[k].double().abs().sum())), (k, tot, ref)
    # comm_rate counts the non-zeros of a ReLU output: a pre-activation within rounding of 0 may land on either side
    # (the convolutions before it sum in a different order than the CPU's), so a couple of cells out of ~5e5 may differ
    assert abs(out["comm_rate"] - float(fx["comm_rate"])) <= max(2.0, 1e-5 * float(fx["comm_rate"]))
    assert out["mask"] == 0
    assert set(out.keys()) == {"psm", "rm", "obj", "mask", "comm_rate"}
    o2 = model(dd)
    assert torch.equal(o2["psm"], out["psm"]) and o2["comm_rate"] == out["comm_rate"]
    if "full" in name:   # default 704 x 200 grid (563 200-input MLP layers): the reference's outputs are the check
        return
    # stage by stage against the oracle (same weights): warp, policy network, keys
    orc = {}
    with torch.no_grad():
        w2.when2com_forward(dd, sd, args, trace=orc)
    assert_close(tr["warped0"].cpu(), orc["warped0"], 1e-4, 1e-5, "warped")
    assert_close(tr["policy0"].cpu(), orc["policy0"], 3e-4, 3e-4, "policy map")
    assert_close(tr["keys0"].cpu(), orc["keys0"], 3e-4, 3e-4, "keys")
    assert_close(tr["query0"].cpu(), orc["query0"], 3e-4, 3e-4, "query")


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,act", [(1, 32, 256, 0), (3, 256, 11264, 1), (8, 128, 256, 1), (11, 40, 4100, 0), (2, 7, 8, 1)])
def test_gpu_linear_rows(m, n, k, act):
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m * 1000 + n)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act:
        ref = ref.relu()
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = torch.empty((m, n), device="cuda")
    need = lib.av2x_linear_rows_workspace_bytes(m, n, k)
    ws = torch.empty(max(need // 4, 1), device="cuda")
    P = lambda t: c_void_p(t.data_ptr())
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.av2x_linear_rows(P(xd), P(wd), P(bd), m, n, k, act, P(y), P(ws), ws.numel() * 4, st), "linear_rows")
    assert_close(y.cpu(), ref.float(), 1e-5, 1e-5, "linear_rows")
    y2 = torch.empty_like(y)
    _lib.check(lib.av2x_linear_rows(P(xd), P(wd), P(bd), m, n, k, act, P(y2), P(ws), ws.numel() * 4, st), "linear_rows")
    assert torch.equal(y, y2), "split-K partials are reduced in a fixed order"
    assert lib.av2x_linear_rows(P(xd), P(wd), P(bd), m, n, k, act, P(y), P(ws), 0, st) != 0     # workspace too small
    assert lib.av2x_linear_rows(P(xd), P(wd), P(bd), m, n, k + 2, act, P(y), P(ws), ws.numel() * 4, st) != 0   # k % 4


@pytest.mark.gpu
def test_gpu_when2com_fuse_kernel():
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    n, ks, e = 5, 256, 4 * 1237
    keys, q = torch.randn(n, ks, generator=g) / 8, torch.randn(ks, generator=g) / 2
    maps = [torch.randn(e, generator=g).cuda() for _ in range(n)]       # separate allocations on purpose
    p = torch.softmax(keys.double() @ q.double(), 0)
    ref = sum(p[j] * maps[j].cpu().double() for j in range(n))
    out, coef = torch.empty(e, device="cuda"), torch.empty(n, device="cuda")
    arr = (c_void_p * n)(*[t.data_ptr() for t in maps])
    P = lambda t: c_void_p(t.data_ptr())
    kd, qd = keys.cuda(), q.cuda()
    _lib.check(lib.av2x_when2com_fuse(P(kd), P(qd), n, ks, arr, e, P(out), P(coef), c_void_p(torch.cuda.current_stream().cuda_stream)), "fuse")
    assert_close(coef.cpu(), p.float(), 1e-5, 1e-6, "coef")
    assert_close(out.cpu(), ref.float(), 1e-5, 1e-5, "fused")
