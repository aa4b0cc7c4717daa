"""GPU end-to-end parity of the Where2Comm forward (through the drop-in module and the C-ABI)
against (a) the golden vectors captured from the real reference and (b) the CPU oracle run on
the same seeded inputs, at the small test grid and at the full AirV2X grid (BASELINE config 2).

fp tolerance (fp32 end to end, ~25 conv layers deep): 2e-4 relative + 2e-4 absolute on the head
outputs whose magnitude is O(1..10).  Mask: bit-exact except cells within 1e-6 of the threshold.
"""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import where2comm_oracle as orc
from tests.helpers import assert_close, case_from_fixture, load_fixture, sample

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2e-4, 2e-4
MAX_FLIPS = 8      # communication-mask cells allowed to sit on the other side of the threshold, each within 1e-6 of it


def _run(name, throughput=False):
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture(name)
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    model.engine().throughput_mode = throughput
    trace = {}
    out = model.engine().forward(dd, trace=trace, sync_comm_rate=True)
    torch.cuda.synchronize()
    return fx, args, sd, dd, out, trace, model


def _mask_ok(got, ref, cmap, what):
    near = np.abs(np.asarray(cmap, np.float64) - 0.01) < 1e-6
    bad = (np.asarray(got) != np.asarray(ref)) & ~near
    assert not bad.any(), f"{what}: {int(bad.sum())} mask cells differ away from the threshold"
    return int(((np.asarray(got) != np.asarray(ref)) & near).sum())


# "+T": the engine in throughput mode (what FramePipeline / the headline's three frames in flight run: the 128- and 256-channel backbone
# layers on the F(4x4,3x3) class, engine.wino4_rule) against the same goldens at the same tolerances
@pytest.mark.parametrize("name", ["w2c_small_n3", "w2c_small_n1", "w2c_full_n2", "w2c_full_n4", "w2c_full_n8", "w2c_full_n2+T", "w2c_full_n4+T", "w2c_full_n8+T"])   # full_n2: BASELINE configs[0]; full_n8: north_star's target workload (8 agents)
def test_forward_matches_reference_golden(name):
    throughput = name.endswith("+T")
    name = name[:-2] if throughput else name
    fx, args, sd, dd, out, tr, model = _run(name, throughput)
    if throughput:      # the mode really changes the class of the block-1 / block-2 layers at this grid
        eng = model.engine()
        assert eng.wino4_rule(eng.blocks[1][1], 1, 50, 176) and eng.wino4_rule(eng.blocks[2][1], 1, 25, 88)
    s, bs = int(fx["sample_stride"]), int(fx["big_stride"])
    assert int(out["comm_rate"]) == int(fx["comm_rate"])            # integer-exact scatter bookkeeping
    flips = _mask_ok(sample(tr["comm_mask"], s), fx["comm_mask"], fx["comm_map"], "comm_mask")
    for i in range(3):
        assert_close(sample(tr[f"block{i}"], bs), fx[f"block{i}"], RTOL, ATOL, f"block{i}")
    assert_close(sample(tr["spatial_features_2d"], bs), fx["spatial_features_2d"], RTOL, ATOL, "sf2d")
    assert_close(sample(tr["shrink"], bs), fx["shrink"], RTOL, ATOL, "shrink")
    assert_close(sample(tr["psm_single"], s), fx["psm_single"], RTOL, ATOL, "psm_single")
    assert_close(sample(tr["comm_map"], s), fx["comm_map"], 1e-4, 1e-7, "comm_map")
    print(f"[{name}] communication-mask cells flipped within 1e-6 of the threshold: {flips}")
    assert flips <= MAX_FLIPS, f"{flips} mask cells flipped at the threshold (allowed: {MAX_FLIPS})"
    # everything downstream of the mask, UNCONDITIONALLY: the oracle (bit-equal to the reference on equal masks, asserted when the
    # fixture was made) is run with the DEVICE's mask replayed, and every element of the fused maps and the heads is compared
    otr = {}
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args, trace=otr, comm_mask=tr["comm_mask"].cpu())
    for i in range(3):
        assert_close(tr[f"fused{i}"].cpu(), otr[f"fused{i}"], RTOL, ATOL, f"fused{i} (device mask replayed)")
    for k in ("psm", "rm", "obj"):
        assert list(out[k].shape) == list(fx[k + "_shape"])
        assert_close(out[k].cpu(), ref[k], RTOL, ATOL, k + " (device mask replayed)")
    if flips == 0:   # identical masks: additionally the reference's own stored samples and sums
        for i in range(3):
            assert_close(sample(tr[f"fused{i}"][0], s), fx[f"fused{i}"], RTOL, ATOL, f"fused{i}")
        for k in ("psm", "rm", "obj"):
            assert_close(sample(out[k], s), fx[k], RTOL, ATOL, k)
            assert abs(out[k].double().sum().item() - float(fx[k + "_sum"])) <= 2e-4 * float(fx[k + "_abssum"])
        # (throughput mode: the F(4x4) class of more layers may move a cell that sits on the threshold -- full_n2: ONE of 70 400, not in the
        # strided sample above; the communication rate then differs by that many cells)
        cells = tr["comm_mask"].numel() // tr["comm_mask"].shape[0]
        assert abs(float(out["com"]) - float(fx["com"])) < (1e-6 if not throughput else (MAX_FLIPS + 0.5) / cells)
    else:            # the rate counts the mask's ones: it moves by exactly the flipped cells
        n_cells = float(np.prod(fx["comm_mask_shape"])) if "comm_mask_shape" in fx else None
        if n_cells:
            assert abs(float(out["com"]) - float(fx["com"])) <= flips / n_cells * len(fx["types"]) + 1e-6


def test_forward_matches_oracle_full_grid_every_element():
    """Full AirV2X grid, 4 agents x 8192 points: every output element against the CPU oracle."""
    fx, args, sd, dd, out, tr, model = _run("w2c_full_n4")
    otr = {}
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args, trace=otr)
    near = (otr["comm_map"] - 0.01).abs() < 1e-6
    differs = (tr["comm_mask"].cpu() != otr["comm_mask"])
    assert not (differs & ~near).any()
    assert_close(tr["spatial_features"].cpu(), otr["spatial_features"], 1e-4, 1e-5, "canvas")
    assert torch.equal(tr["spatial_features"].cpu() != 0, otr["spatial_features"] != 0)
    flips = int(differs.sum())
    print(f"[w2c_full_n4 vs oracle] mask cells flipped within 1e-6 of the threshold: {flips}")
    assert flips <= MAX_FLIPS
    if flips:   # replay the device's mask: the comparison below never depends on a threshold coincidence
        with torch.no_grad():
            ref = orc.where2com_forward(dd, sd, args, comm_mask=tr["comm_mask"].cpu())
    for k in ("psm", "rm", "obj"):
        assert_close(out[k].cpu(), ref[k], RTOL, ATOL, k)
    if not flips:
        assert abs(float(out["com"]) - float(ref["com"])) < 1e-6
    assert int(out["comm_rate"]) == ref["comm_rate"]


def test_module_contract():
    """state_dict keys, output keys/types, weight refresh after load_state_dict."""
    from airv2x_perception_amd import synth
    fx, args, sd, dd, out, tr, model = _run("w2c_small_n3")
    assert list(model.state_dict().keys()) == [str(k) for k in fx["spec_keys"]]
    assert set(out.keys()) == {"psm", "rm", "obj", "mask", "com", "comm_rate"}
    assert out["mask"] == 0 and isinstance(out["comm_rate"], int) and out["com"].dim() == 0
    o1 = model(dd)
    assert torch.equal(o1["psm"], out["psm"])                         # deterministic
    sd2 = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=5)
    model.load_state_dict(sd2)
    o2 = model(dd)
    assert not torch.equal(o2["psm"], out["psm"])                     # packed weights were refreshed
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd2, args)
    assert_close(o2["rm"].cpu(), ref["rm"], 5e-4, 5e-4, "rm after reload")
    ot = model.train()(dd)                                            # train mode: the autograd graph (tests/test_gpu_train.py)
    assert ot["psm"].requires_grad and ot["psm"].shape == out["psm"].shape
    model.eval()


def test_batch_of_two_frames_equals_two_single_frames():
    """B = 2 (the reference's collate layout, different agent mixes per sample): the fusion is per sample, so the batched
    forward must reproduce the two single-frame forwards bit for bit; comm statistics are per-batch sums / means."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd3, voxd, types = case_from_fixture(fx)           # sample 0: vehicle, rsu, drone
    dd2 = synth.build_data_dict([voxd[0], voxd[2]], ["vehicle", "drone"], max_cav_num=args["max_cav_num"])   # sample 1
    both = synth.merge_frames([dd3, dd2])
    assert both["record_len"].tolist() == [3, 2] and both["vehicle"]["record_len"].tolist() == [1, 1]
    assert both["rsu"]["batch_idxs"] == [0] and both["drone"]["batch_idxs"] == [0, 1]
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    o3 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd3, sync_comm_rate=True).items()}
    o2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd2, sync_comm_rate=True).items()}
    ob = eng.forward(both, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert ob[k].shape[0] == 2
        assert torch.equal(ob[k][0:1], o3[k]) and torch.equal(ob[k][1:2], o2[k]), k
    assert ob["comm_rate"] == o3["comm_rate"] + o2["comm_rate"]
    assert abs(float(ob["com"]) - (float(o3["com"]) + float(o2["com"])) / 2) < 1e-6      # where2comm_fuse.py:147 mean over B


def test_full_grid_permutation_of_non_ego_agents_is_invariant():
    """Size-independent property at the BASELINE grid (704 x 200, 5 agents x 8192 points): the ego row of the per-pixel
    attention (where2comm_fuse.py:152-164) is a softmax-weighted SUM over agents, so permuting the non-ego agents of a
    type must leave psm / rm / obj unchanged up to fp32 summation order, and comm statistics exactly unchanged."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.voxelizer import voxelize_points
    hy = synth.default_hypes()
    args, pp = hy["model"]["args"], hy["preprocess"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=4)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    types = ["vehicle", "vehicle", "vehicle", "rsu", "rsu"]
    vox_dev = [voxelize_points(torch.from_numpy(synth.synthetic_cloud(10 + i, 8192)).cuda(), pp["cav_lidar_range"],
                               pp["args"]["voxel_size"], 32, pp["args"]["max_voxel_test"], range_filter=True) for i in range(5)]
    a = model(synth.build_data_dict_device(vox_dev, types, "cuda", max_cav_num=args["max_cav_num"]))
    a = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.items()}
    perm = [0, 2, 1, 4, 3]          # ego stays first; vehicles 1<->2, rsus 0<->1
    b = model(synth.build_data_dict_device([vox_dev[i] for i in perm], types, "cuda", max_cav_num=args["max_cav_num"]))
    for k in ("psm", "rm", "obj"):
        assert_close(b[k].cpu(), a[k].cpu(), 1e-4, 1e-4, f"permutation invariance {k}")
    assert int(b["comm_rate"]) == int(a["comm_rate"]) and abs(float(b["com"]) - float(a["com"])) < 1e-7
    # and the ego DOES matter: a different ego changes the result
    c = model(synth.build_data_dict_device([vox_dev[i] for i in [1, 0, 2, 3, 4]], types, "cuda", max_cav_num=args["max_cav_num"]))
    assert float((c["psm"] - a["psm"]).abs().max()) > 1e-2


@pytest.mark.gpu
def test_hipgraph_replay_equals_eager_launches():
    """engine.use_graph: everything after the scatter is captured once per frame layout and replayed; results must equal
    the eager launches bit for bit, for repeated frames and after new inputs were scattered into the same canvas."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    keep = lambda o: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}
    ref = keep(eng.forward(dd, sync_comm_rate=True))
    dd2 = synth.build_data_dict([voxd[1], voxd[0], voxd[2]], types, max_cav_num=args["max_cav_num"])   # other clouds, same layout
    ref2 = keep(eng.forward(dd2, sync_comm_rate=True))
    eng.use_graph = True
    try:
        for want, inp in ((ref, dd), (ref2, dd2), (ref, dd)):
            out = eng.forward(inp, sync_comm_rate=True)
            assert eng.graph_active()
            for k in ("psm", "rm", "obj"):
                assert torch.equal(out[k], want[k]), k
            assert out["comm_rate"] == want["comm_rate"] and float(out["com"]) == float(want["com"])
    finally:
        eng.use_graph = False


@pytest.mark.gpu
def test_agent_streams_schedule_equals_single_stream():
    """engine.agent_streams = 2: the per-agent part of a B = 1 frame runs as two agent groups on two HIP streams (a
    schedule measured as no faster and left off by default); results must not change."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd, _, _ = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.forward(dd, sync_comm_rate=True).items()}
    eng.agent_streams = 2
    try:
        out = eng.forward(dd, sync_comm_rate=True)
        torch.cuda.synchronize()
    finally:
        eng.agent_streams = 1
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"] and float(out["com"]) == float(ref["com"])


@pytest.mark.gpu
def test_fully_connected_communication_matches_oracle():
    """where2com_fusion.fully = true (where2comm_fuse.py:222-223): no confidence mask, communication rate 1."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from oracle import where2comm_oracle as orc
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd, _, _ = case_from_fixture(fx)
    args = synth.clone_hypes(hy)["model"]["args"]
    args["where2com_fusion"]["fully"] = True
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    out = model(dd)
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert_close(out[k].cpu(), ref[k], 2e-4, 2e-4, k)
    assert int(out["com"]) == int(ref["com"]) == 1
    masked = orc.where2com_forward(dd, sd, synth.clone_hypes(hy)["model"]["args"])
    assert float((masked["psm"] - ref["psm"]).abs().max()) > 1e-3        # the mask does change the result on this frame


@pytest.mark.parametrize("mode", ["x3", "f32"])
@pytest.mark.parametrize("name", ["w2c_small_n3", "w2c_full_n4"])
def test_default_forward_does_not_depend_on_the_tuning_outcome(name, mode, monkeypatch):
    """The module default ("rule" stream-K + autotuned tiles) must give the same bits whatever the autotuner picks: every
    candidate of a numerics class is bit-identical and the stream-K schedule is a function of the layer shape.  Engine A
    tunes by wall-clock; engine B is forced to take the LAST candidate of every class, engine C the first; a third
    forward goes through FramePipeline.  All equal bit for bit.  mode "f32": the fp32-input MFMA kernels (AV2X_X3=0), where nearly every
    layer goes through the tuner; mode "x3" (the default): the split-3 kernels are picked by shape rules, the tuner only sees what is left."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.engine import FramePipeline, Where2ComEngine
    monkeypatch.setenv("AV2X_TUNE_CACHE", "0")
    monkeypatch.setattr(Where2ComEngine, "_tune_disk", None)
    fx = load_fixture(name)
    hy, args, sd, dd, _, _ = case_from_fixture(fx)
    outs, picks = [], []
    for which in ("timed", "last", "first"):
        model = Airv2xWhere2com(args)
        model.load_state_dict(sd)
        model = model.to("cuda").eval()
        eng = model.engine()
        assert eng.stream_k == "rule"
        assert eng.wino_x3 and eng.x3p, "x3 is the default product mode"
        if mode == "f32":
            eng.wino_x3 = eng.x3p = False
        if which != "timed":
            def forced(d, x, L, out, skc=False, key=None, _eng=eng, _w=which):
                c = _eng._candidates(d, L, skc)
                bm, bn, g = c[-1] if _w == "last" else c[0]
                return (bm << 16) | bn, g
            eng._tune = forced
        out = model(dd)
        torch.cuda.synchronize()
        outs.append({k: out[k].clone() for k in ("psm", "rm", "obj")})
        picks.append(dict(eng.tile_cache))
        if which == "timed":
            # FramePipeline puts the engine in throughput mode (more layers on the F(4x4,3x3) class at the full grid: engine.wino4_rule):
            # a pipelined frame equals the single-stream frame of the engine in THAT mode, bit for bit
            pipe = FramePipeline(eng, 2)
            assert pipe.throughput_mode and not eng.throughput_mode      # the mode belongs to the pipeline's frames, not to the caller's engine
            with eng.frame_mode(True, False):
                want_t = {k: v.clone() for k, v in model(dd).items() if k in ("psm", "rm", "obj")}
            again = model(dd)                                            # ... whose own direct calls keep the bits they had before
            for k in ("psm", "rm", "obj"):
                assert torch.equal(again[k], outs[-1][k]), k
            po, ev = pipe.submit(dd)
            ev.synchronize()
            for k in ("psm", "rm", "obj"):
                assert torch.equal(po[k], want_t[k]), k
                assert_close(want_t[k].cpu(), outs[0][k].cpu().numpy(), 1e-4, 1e-4, f"throughput vs latency mode {k}")
    if mode == "f32":
        assert picks[1] != picks[2]                                        # the forced engines really ran different kernels
        assert any(len(k) > 6 and k[6] == "rule" for k in picks[0]) or name == "w2c_small_n3"   # full grid: some layers take the stream-K rule
        assert any(k[0] == "wino" for k in picks[0])       # ... and the Winograd layers went through the tuner as well (bit-identical tilings)
    for o in outs[1:]:
        for k in ("psm", "rm", "obj"):
            assert torch.equal(o[k], outs[0][k]), k


def test_workspace_pool_is_bounded_when_the_frame_layout_keeps_changing():
    """A scenario stream changes its agent count every few frames; the workspace pool must not grow with every new
    layout.  With a limit between one layout and all three the engine evicts least-recently-used buffers (never one of the running frame)
    and every frame still equals the unbounded engine's result bit for bit."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    layouts = [synth.build_data_dict(voxd[:k], types[:k], max_cav_num=args["max_cav_num"]) for k in (1, 2, 3)]
    ref_model = Airv2xWhere2com(args)
    ref_model.load_state_dict(sd)
    ref_model = ref_model.to("cuda").eval()
    want = [{k: ref_model(d)[k].clone() for k in ("psm", "rm", "obj")} for d in layouts]
    unbounded = ref_model.engine()._ws_bytes
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    model(layouts[2])
    one = eng._ws_bytes                      # the largest layout alone
    assert unbounded > one
    eng.ws_limit = one + (unbounded - one) // 2
    assert unbounded > eng.ws_limit          # all three layouts together do not fit
    peak = 0
    for it in range(12):
        k = (it * 2) % 3
        out = model(layouts[k])
        for key in ("psm", "rm", "obj"):
            assert torch.equal(out[key], want[k][key]), (it, key)
        peak = max(peak, eng._ws_bytes)
    assert peak <= eng.ws_limit + one        # bounded: the limit plus at most the frame in progress


def test_comm_rate_counted_in_the_scatter_equals_the_read_back_count():
    """comm_rate = spatial_features.count_nonzero() (airv2x_where2com.py:122): the LiDAR-only frame counts while it scatters
    (av2x_pillar_vfe_scatter_count); with the fold switched off the same engine counts by reading the canvas back.  Same integer, and the
    one the reference's own model produced (fixture)."""
    fx, args, sd, dd, out, tr, model = _run("w2c_small_n3")
    eng = model.engine()
    assert eng.FOLD_COUNT
    a = eng.forward(dd, sync_comm_rate=True)["comm_rate"]
    eng.FOLD_COUNT = False
    try:
        b = eng.forward(dd, sync_comm_rate=True)["comm_rate"]
    finally:
        eng.FOLD_COUNT = True
    c = eng.forward(dd, sync_comm_rate=True)["comm_rate"]
    assert a == b == c == int(fx["comm_rate"])


def test_sparse_first_convolution_of_the_frame_equals_the_dense_one():
    """The first backbone convolution of a LiDAR-only frame gathers the occupied taps of the scattered canvas (av2x_conv3x3s2_sparse);
    switched off, the same engine runs the dense split-3 GEMM on the whole canvas.  Same frame: block 0 and the heads within fp32 rounding
    of each other, the communication mask and comm_rate identical."""
    fx, args, sd, dd, out, tr, model = _run("w2c_small_n3")
    eng = model.engine()
    assert eng.SPARSE_CONV0
    eng.SPARSE_CONV0 = False
    try:
        tr2 = {}
        out2 = eng.forward(dd, trace=tr2, sync_comm_rate=True)
    finally:
        eng.SPARSE_CONV0 = True
    assert out2["comm_rate"] == out["comm_rate"]
    assert torch.equal(tr2["spatial_features"], tr["spatial_features"])
    assert_close(tr2["block0"].cpu(), tr["block0"].cpu().numpy(), 1e-5, 1e-5, "block0 dense vs sparse first conv")
    for k in ("psm", "rm"):
        assert_close(out2[k].cpu(), out[k].cpu().numpy(), 1e-4, 1e-4, k)
