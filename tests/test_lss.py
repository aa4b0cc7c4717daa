"""Camera lift-splat (BASELINE configs[4]): create_frustum / get_geometry / voxel_pooling / fuse_bev.
Goldens = the reference's own methods (airv2x_encoder.py:94-275) run by tools/gen_golden.py `lss` on seeded camera rigs:
a small rig with every tensor stored, and configs[4]'s shapes (360x640 images / 8 -> 45x80, 48 LID depth bins for the
vehicle cameras, 144 UD bins for the drone's, 704x200 BEV grid; strided samples + sums + per-cell point counts).
The EfficientNet / ResNet trunk and BevEncode are unpinned and not built."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import lss_oracle as lo
from tests.helpers import assert_close, load_fixture

NAMES = ["lss_small", "lss_small_dense", "lss_cfg4_vehicle", "lss_cfg4_drone"]


def _case(fx):
    at = str(fx["agent_type"])
    ca = synth.cam_args(at, tuple(int(v) for v in fx["final_dim"]), tuple(float(v) for v in fx["xy"]))
    B, N = int(fx["B"]), int(fx["N"])
    rig = synth.camera_rig(int(fx["seed"]), B, N, tuple(int(v) for v in fx["final_dim"]), drone=(at == "drone"))
    D, fH, fW = fx["frustum"].shape[:3]
    x = synth.lifted_features(int(fx["seed"]) + 1, B, N, D, fH, fW, ca["img_features"], one_hot=bool(fx["one_hot"]))
    return ca, rig, x, B, N


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(name):
    fx = load_fixture(name)
    ca, rig, x, B, N = _case(fx)
    gc, s = ca["grid_conf"], int(fx["stride"])
    fr = lo.create_frustum(gc, ca["data_aug_conf"], ca["img_downsample"])
    assert np.array_equal(fr.numpy(), fx["frustum"])                       # host constants: bit-exact
    geom = lo.get_geometry(fr, *rig)
    # bit-identical on the machine that generated the golden; torch's batched 3x3 matmul takes different (FMA / non-FMA)
    # kernels on other CPUs -> last-bit slack, and with it a handful of points within an ulp of a voxel face
    assert_close(geom[:, :, ::max(1, s // 2), ::s, ::s].numpy(), fx["geom"], 2e-6, 2e-5, "geometry")
    dx, bx, nx = lo.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
    _, kept = lo.voxel_indices(geom, dx, bx, nx, B)
    assert abs(int(kept.sum()) - int(fx["kept_count"])) <= 4
    bev = lo.voxel_pooling(geom, x, dx, bx, nx)
    tol = 1e-4 * float(np.abs(fx["bev"]).max())
    exact = lo.voxel_pooling_exact(geom, x, dx, bx, nx)
    flipped = (np.abs(exact[..., ::s, ::s].float().numpy() - fx["bev_exact"]) > 1e-5 * (1 + np.abs(fx["bev_exact"]))).any(1, keepdims=True)
    assert int(flipped.sum()) <= 8                                          # cells that gained / lost a boundary point
    ok = ~np.broadcast_to(flipped, fx["bev"].shape)
    assert_close(bev[..., ::s, ::s].numpy()[ok], fx["bev"][ok], 0, tol, "bev")   # argsort order of equal ranks is not defined
    assert torch.equal(lo.fuse_bev([bev, 3 * bev]), (bev + 3 * bev) / 2) or torch.allclose(lo.fuse_bev([bev, 3 * bev]), 2 * bev)


def test_host_mirror_constants_equal_the_oracle():
    from airv2x_perception_amd.opencood_iface import lss
    for at in ("vehicle", "rsu", "drone"):
        ca = synth.cam_args(at)
        gc = ca["grid_conf"]
        assert torch.equal(lss.create_frustum(gc, ca["data_aug_conf"], 8), lo.create_frustum(gc, ca["data_aug_conf"], 8))
        for a, b in zip(lss.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"]), lo.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])):
            assert torch.equal(a, b)
    assert lss.create_frustum(synth.cam_args("drone")["grid_conf"], synth.cam_args("drone")["data_aug_conf"], 8).shape == (144, 45, 80, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_lift_splat_matches_reference_golden(name):
    from airv2x_perception_amd.opencood_iface.lss import LiftSplat
    fx = load_fixture(name)
    ca, rig, x, B, N = _case(fx)
    s = int(fx["stride"])
    ls = LiftSplat(ca, "cuda")
    assert np.array_equal(ls.frustum_host.numpy(), fx["frustum"])
    # geometry: the same fp32 operation order as the reference (torch's 3x3 matmul may fuse a multiply-add: last-bit slack)
    geom = ls.get_geometry(*rig)
    g = geom[:, :, ::max(1, s // 2), ::s, ::s].cpu().numpy()
    assert_close(g, fx["geom"], 2e-6, 2e-5, "geometry")
    # voxel indices: pool an all-ones feature map -> every cell holds its point count; integers, so any index
    # disagreement shows; allow the handful of points that sit within an ulp of a voxel face
    ones = torch.ones_like(x)
    cnt = ls.voxel_pooling(ones, *rig)[..., 0].cpu().numpy()               # (B, ny, nx): nz = 1
    ref_cnt = fx["cell_counts"][:, 0]
    got_cnt = np.rint(cnt[:, ::s, ::s]).astype(np.int64)
    assert np.abs(cnt - np.rint(cnt)).max() < 1e-6
    assert int((got_cnt != ref_cnt).sum()) <= 4, int((got_cnt != ref_cnt).sum())
    assert abs(int(np.rint(cnt).sum()) - int(fx["kept_count"])) <= 4
    # pooled features: NHWC (B, ny, nx, nz*C) vs the reference's (B, nz*C, ny, nx); at least as close to the float64
    # pooling as the reference's own cumsum-difference form
    bev = ls.voxel_pooling(x, *rig).permute(0, 3, 1, 2)
    mine = bev[..., ::s, ::s].cpu().numpy()
    scale = float(np.abs(fx["bev_exact"]).max())
    flips = got_cnt != ref_cnt
    mask = ~np.broadcast_to(flips[:, None], mine.shape)
    err = np.abs(mine - fx["bev_exact"])[mask].max()
    assert err <= max(float(fx["reference_max_err"]), 2e-6 * scale), (err, float(fx["reference_max_err"]))
    assert_close(mine[mask], fx["bev"][mask], 0, 2e-4 * scale, "bev vs reference")
    assert abs(float(bev.double().sum()) - float(fx["bev_exact_sum"])) <= 1e-4 * float(fx["bev_exact_abssum"])
    # bit-reproducible (fixed-point integer atomics): a second run gives identical bits
    assert torch.equal(ls.voxel_pooling(x, *rig), bev.permute(0, 2, 3, 1))
    print(f"[{name}] device pooling vs float64: max err {err:.2e} (reference's own: {float(fx['reference_max_err']):.2e})")


@pytest.mark.gpu
def test_gpu_fuse_bev_is_the_modality_mean():
    from airv2x_perception_amd.opencood_iface.lss import fuse_bev
    g = torch.Generator().manual_seed(1)
    maps = [torch.randn(2, 64, 20, 36, generator=g).cuda() for _ in range(2)]
    assert_close(fuse_bev(maps).cpu(), lo.fuse_bev([m.cpu() for m in maps]), 1e-6, 1e-6, "fuse_bev")
    assert fuse_bev(maps[:1]) is maps[0] or torch.equal(fuse_bev(maps[:1]), maps[0])
