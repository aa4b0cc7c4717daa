"""CPU, world_size 2, gloo: the agent-sharded frame protocol (opencood_iface/sharded.py) —
partitioning, ego placement, the all-gather that replaces the reference's in-process concat and
the stats all-reduce — driven with an oracle-based backend (the HIP engine implements the same
two-method backend interface on the GPU; see test_gpu_sharded.py for that side)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.sharded import ShardedFrame, partition_agents
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc

RNG = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]
TYPES = ["vehicle", "vehicle", "rsu", "drone"]


class OracleBackend:
    """The two-stage backend interface of ShardedFrame on CPU tensors (NCHW, oracle functions)."""

    def __init__(self, sd, args):
        self.sd, self.args = sd, args

    def local_stage(self, dd_local, has_ego, n_pad=None):
        sd, args = self.sd, self.args
        mf = args["modality_fusion"]
        n = 0 if dd_local is None else sum(int(dd_local[t]["record_len"][0]) for t in ("vehicle", "rsu", "drone")
                                            if dd_local[t]["batch_idxs"])
        n_pad = n if n_pad is None else n_pad
        g = args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
        H, W = int(g[1]) // 2, int(g[0]) // 2
        shapes = [(n_pad, 64, H, W), (n_pad, 128, H // 2, W // 2), (n_pad, 256, H // 4, W // 4)]
        if n == 0:   # a rank without agents: all-padding message
            send = torch.zeros(sum(int(np.prod(sh)) for sh in shapes))
            return send, torch.zeros(2, dtype=torch.int64), {"shapes": shapes}
        feats, record_len = orc.extract_features(dd_local, sd, args)
        sf2d, blocks = orc.backbone_forward(feats, sd, mf["base_bev_backbone"])
        s = orc.shrink_conv(sf2d, sd, mf["shrink_header"])
        psm_single = orc.head(s, sd, "cls_head")
        masks, _, maps = orc.communication([psm_single], sd, args["where2com_fusion"]["communication"])
        thr = args["where2com_fusion"]["communication"]["threshold"]
        ones = (maps > thr)
        if not has_ego:
            masks = ones.float()  # no ego on this rank: nothing is forced to 1
        x0 = blocks[0] * masks
        x1 = orc.backbone_block(x0, sd, 1, mf["base_bev_backbone"]["layer_nums"][1])
        x2 = orc.backbone_block(x1, sd, 2, mf["base_bev_backbone"]["layer_nums"][2])
        pad = lambda x: torch.cat([x, x.new_full((n_pad - n,) + tuple(x.shape[1:]), float("nan"))], 0)  # padding must never be read
        send = torch.cat([pad(x0).reshape(-1), pad(x1).reshape(-1), pad(x2).reshape(-1)])
        stats = torch.tensor([int(ones.sum()), int(feats.count_nonzero())], dtype=torch.int64)
        assert [tuple(pad(x).shape) for x in (x0, x1, x2)] == shapes
        return send, stats, {"shapes": shapes}

    def ego_stage(self, recv, stats, meta, world):
        sd, args = self.sd, self.args
        mf = args["modality_fusion"]
        per_rank = recv.numel() // world
        counts = meta.get("counts") or [meta["shapes"][0][0]] * world
        levels = [[], [], []]
        for r in range(world):
            chunk, off = recv[r * per_rank:(r + 1) * per_rank], 0
            for i, shp in enumerate(meta["shapes"]):
                n = int(np.prod(shp))
                levels[i].append(chunk[off:off + n].view(shp)[:counts[r]])
                off += n
        ups = []
        for i in range(3):
            x = torch.cat(levels[i], 0)
            assert not torch.isnan(x).any()
            f = orc.attention_fusion(x).unsqueeze(0)
            ups.append(orc.backbone_deblock(f, sd, i, mf["base_bev_backbone"]["upsample_strides"][i]))
        fs = orc.shrink_conv(torch.cat(ups, 1), sd, mf["shrink_header"])
        n_total = sum(counts)
        H, W = levels[0][0].shape[-2:]
        return {"psm": orc.head(fs, sd, "cls_head"), "rm": orc.head(fs, sd, "reg_head"), "obj": orc.head(fs, sd, "obj_head"),
                "com": stats[0].float() / (n_total * H * W), "comm_rate": int(stats[1])}


def _local_count(dd_local):
    return 0 if dd_local is None else sum(int(dd_local[t]["record_len"][0]) for t in ("vehicle", "rsu", "drone") if dd_local[t]["batch_idxs"])


def _real_agents(recv, meta, world):
    """The gathered (world * n_pad, C, H, W) buffer with the padding slots of an uneven frame dropped (frame order kept)."""
    from airv2x_perception_amd.opencood_iface.sharded import valid_slots
    n_loc, c, h, w = meta["shape"]
    s = recv.view(world * n_loc, c, h, w)
    counts = meta.get("counts")
    if counts is not None and sum(counts) != world * n_loc:
        s = s[valid_slots(counts, n_loc)]
    assert not torch.isnan(s).any()
    return s


class CoBEVTOracleBackend:
    """Same interface for the CoBEVT path: the message is the shrink output, or (compression > 0) the
    NaiveCompressor encoder output that the receiving side decodes (naive_compress.py:12-42)."""

    def __init__(self, sd, args, two_level=False):
        self.sd, self.args, self.two_level = sd, args, two_level

    def _decode(self, recv, meta, world):
        import torch.nn.functional as F
        sd, args = self.sd, self.args
        s = _real_agents(recv, meta, world)
        if args["compression"]:
            for conv, bn in (("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
                s = F.conv2d(s, sd[f"naive_compressor.{conv}.weight"], sd[f"naive_compressor.{conv}.bias"], padding=1)
                s = F.relu(orc._bn(s, sd, f"naive_compressor.{bn}"))
        return s

    def ego_partial(self, recv, stats, meta, world, rank):
        """Second level: this rank fuses its residue-group columns only (sharded.fusion_column_shards)."""
        from airv2x_perception_amd.opencood_iface.sharded import fusion_column_shards
        from oracle import cobevt_oracle as cob
        sd, args = self.sd, self.args
        s = self._decode(recv, meta, world)
        shards = fusion_column_shards(s.shape[-1], args["fax_fusion"]["window_size"], world)
        sc = s[..., shards[rank][0]]
        x, mask = cob.regroup(sc, torch.tensor([sc.shape[0]]), sum(args["max_cav"].values()))
        fused = cob.swap_fusion_encoder(x, mask, sd, args["fax_fusion"])
        heads = torch.cat([orc.head(fused, sd, n) for n in ("cls_head", "reg_head", "obj_head")], 1)
        return heads.reshape(-1).contiguous(), {"shards": shards, "shape": tuple(heads.shape), "W": s.shape[-1]}

    def ego_finish(self, parts, ctx, world):
        nb, nh, H, Wc = ctx["shape"]
        full = torch.zeros(nb, nh, H, ctx["W"])
        per = parts.view(world, nb, nh, H, Wc)
        for r, (cols, valid) in enumerate(ctx["shards"]):
            strip = Wc // 4
            keep = [w2 * strip + j for w2 in range(4) for j in range(valid)]
            full[..., [cols[k] for k in keep]] = per[r][..., keep]
        a = self.args["anchor_number"]
        c = a * self.args["num_class"]
        return {"psm": full[:, :c], "rm": full[:, c:c + 7 * a], "obj": full[:, c + 7 * a:]}

    def local_stage(self, dd_local, has_ego, n_pad=None):
        import torch.nn.functional as F
        sd, args = self.sd, self.args
        n = _local_count(dd_local)
        n_pad = n if n_pad is None else n_pad
        g = args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
        c = args["shrink_header"]["dim"][-1] // (args["compression"] or 1)
        shape = (n_pad, c, int(g[1]) // 2, int(g[0]) // 2)
        if n == 0:   # a rank without agents: an all-padding message that must never be read
            return torch.full((int(np.prod(shape)),), float("nan")), torch.zeros(2, dtype=torch.int64), {"shape": shape}
        feats, _ = orc.extract_features(dd_local, sd, args)
        sf2d, _ = orc.backbone_forward(feats, sd, args["base_bev_backbone"])
        s = orc.shrink_conv(sf2d, sd, args["shrink_header"])
        if args["compression"]:
            s = F.conv2d(s, sd["naive_compressor.encoder.0.weight"], sd["naive_compressor.encoder.0.bias"], padding=1)
            s = F.relu(orc._bn(s, sd, "naive_compressor.encoder.1"))
        s = torch.cat([s, s.new_full((n_pad - n,) + tuple(s.shape[1:]), float("nan"))], 0)
        assert tuple(s.shape) == shape
        return s.reshape(-1), torch.zeros(2, dtype=torch.int64), {"shape": shape}

    def ego_stage(self, recv, stats, meta, world):
        import torch.nn.functional as F
        from oracle import cobevt_oracle as cob
        sd, args = self.sd, self.args
        s = self._decode(recv, meta, world)
        L = sum(args["max_cav"].values())
        x, mask = cob.regroup(s, torch.tensor([s.shape[0]]), L)
        fused = cob.swap_fusion_encoder(x, mask, sd, args["fax_fusion"])
        return {"psm": orc.head(fused, sd, "cls_head"), "rm": orc.head(fused, sd, "reg_head"), "obj": orc.head(fused, sd, "obj_head")}


class When2comOracleBackend:
    """Same interface for the When2com path: the message is the agent's map warped into the ego frame, its key and
    (from the ego's rank) the projected query."""

    def __init__(self, sd, args):
        self.sd, self.args = sd, args

    def local_stage(self, dd_local, has_ego):
        import torch.nn.functional as F
        from oracle import when2com_oracle as w2
        sd, args = self.sd, self.args
        mf, cfg = args["modality_fusion"], args["when2com_fusion"]
        feats, _ = orc.extract_features(dd_local, sd, args)
        sf2d, _ = orc.backbone_forward(feats, sd, mf["base_bev_backbone"])
        s = orc.shrink_conv(sf2d, sd, mf["shrink_header"])
        n, C, H, W = s.shape
        t = w2.normalized_pairwise(dd_local["img_pairwise_t_matrix_collab"], H, W, cfg["voxel_size"][0], cfg["downsample_rate"])
        off = dd_local["shard_rank"] * n
        nb = w2.warp_affine_simple(s, t[0, 0, off:off + n], (H, W))
        qk = w2.policy_net(nb, sd, "fusion_net.query_key_net")
        keys = w2.km_generator(qk, sd, "fusion_net.key_net")
        q = torch.zeros(cfg["key_size"])
        if has_ego:
            query = w2.km_generator(qk[0:1], sd, "fusion_net.query_net")
            q = F.linear(query, sd["fusion_net.attention_net.linear.weight"], sd["fusion_net.attention_net.linear.bias"]).view(-1)
        send = torch.cat([nb.reshape(-1), keys.reshape(-1), q])
        return send, torch.tensor([0, int(s.count_nonzero())], dtype=torch.int64), {"shape": tuple(nb.shape), "ks": cfg["key_size"]}

    def ego_stage(self, recv, stats, meta, world):
        sd = self.sd
        n, C, H, W = meta["shape"]
        ks, per = meta["ks"], recv.numel() // world
        chunks = recv.view(world, per)
        maps = chunks[:, :n * C * H * W].reshape(world * n, C, H, W)
        keys = chunks[:, n * C * H * W:n * C * H * W + n * ks].reshape(world * n, ks)
        q = chunks[0, n * C * H * W + n * ks:]
        coef = torch.softmax(keys @ q, 0)
        fused = (coef.view(-1, 1, 1, 1) * maps).sum(0, keepdim=True)
        return {"psm": orc.head(fused, sd, "cls_head"), "rm": orc.head(fused, sd, "reg_head"), "obj": orc.head(fused, sd, "obj_head"),
                "comm_rate": int(stats[1]) / 1}


def _when2com_frame():
    hy = synth.default_hypes_when2com(RNG)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.when2com_param_spec(args), seed=4)
    _, _, voxd = _frame()
    return args, sd, voxd


def _when2com_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args, sd, voxd = _when2com_frame()
    mine = partition_agents(len(TYPES), world)[rank]
    dd_local = synth.build_data_dict([voxd[i] for i in mine], [TYPES[i] for i in mine])
    dd_local["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(len(TYPES), args["max_cav_num"])
    with torch.no_grad():
        out = ShardedFrame(When2comOracleBackend(sd, args)).forward(dd_local)   # sets shard_rank
    gathered = [torch.empty_like(out["psm"]) for _ in range(world)]
    dist.all_gather(gathered, out["psm"])
    assert all(torch.equal(g, gathered[0]) for g in gathered)                   # every rank finishes the frame
    if rank == 0:
        torch.save(out, result_path)
    dist.destroy_process_group()


def test_when2com_agent_sharded_frame_equals_single_process(tmp_path):
    from oracle import when2com_oracle as w2
    path = str(tmp_path / "out.pt")
    mp.spawn(_when2com_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd = _when2com_frame()
    dd = synth.build_data_dict(voxd, TYPES)
    dd["img_pairwise_t_matrix_collab"] = synth.when2com_pairwise(len(TYPES), args["max_cav_num"])
    with torch.no_grad():
        ref = w2.when2com_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), k
    assert got["comm_rate"] == ref["comm_rate"]


class V2XViTOracleBackend:
    """V2X-ViT: the message is the shrink-header map; second level = column strips of the encoder blocks with ONE
    all-reduce per block (the split-attention mean over the map), then a gather of the head outputs."""

    def __init__(self, sd, args, two_level=False, msg_dtype=torch.float32):
        # msg_dtype bfloat16: the autocast frame's message (engine.msg_dtype on the GPU side): rounded once by the sender, widened exactly
        # by the receiver -- 2 bytes per element through the all-gather
        self.sd, self.args, self.two_level, self.msg_dtype = sd, args, two_level, msg_dtype

    def local_stage(self, dd_local, has_ego, n_pad=None):
        sd, args = self.sd, self.args
        mf = args["modality_fusion"]
        n = _local_count(dd_local)
        n_pad = n if n_pad is None else n_pad
        g = args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
        # NaiveCompressor (airv2x_v2xvit.py:42-44, 122-123): the message is the ENCODER's output, 256 / ratio channels; the receiver decodes
        self.ratio = synth.model_compression(args)
        shape = (n_pad, mf["shrink_header"]["dim"][-1] // (self.ratio or 1), int(g[1]) // 2, int(g[0]) // 2)
        meta = {"shape": shape, "prior": dd_local["prior_encoding"], "scm": dd_local["spatial_correction_matrix"]}
        if n == 0:
            return torch.full((int(np.prod(shape)),), float("nan"), dtype=self.msg_dtype), torch.zeros(2, dtype=torch.int64), meta
        feats, _ = orc.extract_features(dd_local, sd, args)
        sf2d, _ = orc.backbone_forward(feats, sd, mf["base_bev_backbone"])
        s = orc.shrink_conv(sf2d, sd, mf["shrink_header"])
        if self.ratio:
            s = F.conv2d(s, sd["naive_compressor.encoder.0.weight"], sd["naive_compressor.encoder.0.bias"], padding=1)
            s = F.relu(orc._bn(s, sd, "naive_compressor.encoder.1"))
        s = torch.cat([s, s.new_full((n_pad - n,) + tuple(s.shape[1:]), float("nan"))], 0)
        return s.reshape(-1).to(self.msg_dtype), torch.tensor([0, int(feats.count_nonzero())], dtype=torch.int64), meta

    def _tokens(self, recv, meta, world):
        from oracle import cobevt_oracle as cob
        n_loc, c, h, w = meta["shape"]
        assert recv.dtype == self.msg_dtype
        s = _real_agents(recv, meta, world).float()
        if synth.model_compression(self.args):      # the two decoder layers, on the receiving side
            for conv, bn in (("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
                s = F.conv2d(s, self.sd[f"naive_compressor.{conv}.weight"], self.sd[f"naive_compressor.{conv}.bias"], padding=1)
                s = F.relu(orc._bn(s, self.sd, f"naive_compressor.{bn}"))
        x, mask = cob.regroup(s, torch.tensor([s.shape[0]]), self.args["max_cav_num"])
        prior = meta["prior"].unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, h, w)
        return torch.cat([x, prior], dim=2).permute(0, 1, 3, 4, 2).contiguous(), mask

    def _heads(self, fused):
        fused = fused.permute(0, 3, 1, 2).contiguous()
        return torch.cat([orc.head(fused, self.sd, n) for n in ("cls_head", "reg_head", "obj_head")], 1)

    def _split(self, heads, stats):
        a = self.args["anchor_number"]
        c = a * self.args["num_class"]
        return {"psm": heads[:, :c], "rm": heads[:, c:c + 7 * a], "obj": heads[:, c + 7 * a:], "comm_rate": int(stats[1])}

    def ego_stage(self, recv, stats, meta, world):
        from oracle import v2xvit_oracle as vit
        x, mask = self._tokens(recv, meta, world)
        return self._split(self._heads(vit.encoder(x, mask, meta["scm"], self.sd, self.args["transformer"]["encoder"])), stats)

    def can_split(self, meta, world):
        return meta["shape"][-1] % (4 * world) == 0

    def ego_partial(self, recv, stats, meta, world, rank):
        from oracle import v2xvit_oracle as vit
        x, mask = self._tokens(recv, meta, world)
        wc = meta["shape"][-1] // world

        def gap_reduce(gap):
            g = gap.clone()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return g / world
        fused = vit.encoder(x, mask, meta["scm"], self.sd, self.args["transformer"]["encoder"], strip=(rank * wc, wc),
                            gap_reduce=gap_reduce)
        heads = self._heads(fused)
        return heads.reshape(-1).contiguous(), {"shape": tuple(heads.shape), "stats": stats}

    def ego_finish(self, parts, ctx, world):
        nb, nh, H, wc = ctx["shape"]
        full = parts.view(world, nb, nh, H, wc).permute(1, 2, 3, 0, 4).reshape(nb, nh, H, world * wc)
        return self._split(full, ctx["stats"])


def _v2xvit_frame(compression=0):
    hy = synth.default_hypes_v2xvit(RNG)
    args = hy["model"]["args"]
    if compression:
        args["modality_fusion"]["compression"] = args["compression"] = int(compression)
    sd = synth.synthetic_state_dict(synth.v2xvit_param_spec(args), seed=5)
    _, _, voxd = _frame()
    dd = synth.build_data_dict(voxd, TYPES, max_cav_num=args["max_cav_num"])
    scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
    for i in range(1, len(TYPES)):
        scm[0, i] = torch.from_numpy(synth.se2_correction(2.0 * i, 0.6 * i, -0.3 * i))
    dd["spatial_correction_matrix"] = scm
    return args, sd, voxd, dd


def _v2xvit_worker(rank, world, port, result_path, two_level, msg_dtype=torch.float32, compression=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args, sd, voxd, dd = _v2xvit_frame(compression)
    mine = partition_agents(len(TYPES), world)[rank]
    dd_local = synth.build_data_dict([voxd[i] for i in mine], [TYPES[i] for i in mine], max_cav_num=args["max_cav_num"])
    for k in ("prior_encoding", "spatial_correction_matrix"):      # frame-level metadata of all agents
        dd_local[k] = dd[k]
    with torch.no_grad():
        frame = ShardedFrame(V2XViTOracleBackend(sd, args, two_level, msg_dtype))
        out = frame.forward(dd_local)
    if rank == 0:
        out["message_bytes"] = frame.last_exchange["message_bytes"]
        torch.save(out, result_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("two_level,ratio", [(False, 2), (True, 4)])
def test_v2xvit_compressed_message_sharded_frame_equals_single_process(tmp_path, two_level, ratio):
    """V2X-ViT with a NaiveCompressor (modality_fusion.compression > 0, ratio = args["compression"]) in the agent-sharded frame: the message
    is the encoder's 256 / ratio-channel output -- the all-gather payload shrinks ratio x (36.0 -> 36.0 / ratio MB per agent at the default
    grid) -- and the receiver runs the decoder; outputs equal the single-process oracle of the same model."""
    from oracle import v2xvit_oracle as vit
    path = str(tmp_path / "out.pt")
    mp.spawn(_v2xvit_worker, args=(2, _free_port(), path, two_level, torch.float32, ratio), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd, dd = _v2xvit_frame(ratio)
    g = args["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
    agents_per_rank = max(len(p) for p in partition_agents(len(TYPES), 2))
    assert got["message_bytes"] == agents_per_rank * (256 // ratio) * (int(g[1]) // 2) * (int(g[0]) // 2) * 4      # ratio x fewer bytes per agent
    with torch.no_grad():
        ref = vit.v2xvit_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), k
    assert got["comm_rate"] == ref["comm_rate"]


@pytest.mark.parametrize("two_level", [False, True])
def test_v2xvit_agent_sharded_frame_equals_single_process(tmp_path, two_level):
    from oracle import v2xvit_oracle as vit
    path = str(tmp_path / "out.pt")
    mp.spawn(_v2xvit_worker, args=(2, _free_port(), path, two_level), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd, dd = _v2xvit_frame()
    with torch.no_grad():
        ref = vit.v2xvit_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), k
    assert got["comm_rate"] == ref["comm_rate"]


@pytest.mark.parametrize("two_level", [False, True])
def test_v2xvit_bf16_message_sharded_frame_equals_single_process(tmp_path, two_level):
    """The autocast frame's message is bf16 (18.0 MB per agent at the default grid instead of 36.0): two ranks exchanging bf16 over the
    all-gather give the bits of ONE process that rounds the shrink-header output the same way (world 1, same backend)."""
    path = str(tmp_path / "out.pt")
    mp.spawn(_v2xvit_worker, args=(2, _free_port(), path, two_level, torch.bfloat16), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd, dd = _v2xvit_frame()
    with torch.no_grad():
        one = ShardedFrame(V2XViTOracleBackend(sd, args, False, torch.bfloat16)).forward(dd)
        exact = ShardedFrame(V2XViTOracleBackend(sd, args, False)).forward(dd)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], one[k], rtol=1e-4, atol=1e-4), k          # strips: summation order of the split-attention mean
        if not two_level:
            assert torch.equal(got[k], one[k]), k
        assert not torch.equal(one[k], exact[k])                                  # the rounding is real ...
        assert float((one[k] - exact[k]).abs().max()) <= 2e-2 * float(exact[k].abs().max())   # ... and small
    assert got["comm_rate"] == one["comm_rate"]


def _cobevt_frame(compression):
    hy = synth.default_hypes_cobevt(RNG, compression=compression)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.cobevt_param_spec(args), seed=3)
    _, _, voxd = _frame()
    return args, sd, voxd


def _cobevt_worker(rank, world, port, result_path, compression, two_level=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args, sd, voxd = _cobevt_frame(compression)
    mine = partition_agents(len(TYPES), world)[rank]
    dd_local = synth.build_data_dict([voxd[i] for i in mine], [TYPES[i] for i in mine])
    with torch.no_grad():
        out = ShardedFrame(CoBEVTOracleBackend(sd, args, two_level)).forward(dd_local)
    if rank == 0:
        torch.save(out, result_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("compression,two_level", [(0, False), (4, False), (0, True)])
def test_cobevt_agent_sharded_frame_equals_single_process(tmp_path, compression, two_level):
    """two_level: besides the agents, the FUSION is split over the ranks by residue-group columns (no exchange between the
    window and grid halves of a block) and the head outputs are all-gathered."""
    from oracle import cobevt_oracle as cob
    path = str(tmp_path / "out.pt")
    mp.spawn(_cobevt_worker, args=(2, _free_port(), path, compression, two_level), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd = _cobevt_frame(compression)
    dd = synth.build_data_dict(voxd, TYPES)
    with torch.no_grad():
        ref = cob.cobevt_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), k


def _frame():
    hy = synth.default_hypes(RNG)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=0)
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, 400, RNG), RNG), RNG, [0.4, 0.4, 4.0])
            for i in range(len(TYPES))]
    return args, sd, voxd


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args, sd, voxd = _frame()
    mine = partition_agents(len(TYPES), world)[rank]
    dd_local = synth.build_data_dict([voxd[i] for i in mine], [TYPES[i] for i in mine])
    with torch.no_grad():
        out = ShardedFrame(OracleBackend(sd, args)).forward(dd_local)
    # every rank ends with the same fused result (SPMD)
    gathered = [torch.empty_like(out["psm"]) for _ in range(world)]
    dist.all_gather(gathered, out["psm"])
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    if rank == 0:
        torch.save({k: out[k] for k in ("psm", "rm", "obj", "com", "comm_rate")}, result_path)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_agent_sharded_frame_equals_single_process(tmp_path):
    path = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = torch.load(path)
    args, sd, voxd = _frame()
    dd = synth.build_data_dict(voxd, TYPES)
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args)
    for k in ("psm", "rm", "obj"):
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), k
    assert got["comm_rate"] == ref["comm_rate"]
    assert abs(float(got["com"]) - float(ref["com"])) < 1e-6


def _uneven_worker(rank, world, port, result_path, n_agents, rotate, gather=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AV2X_SHARD_GATHER"] = "1" if gather else "0"   # rotating ego stage: gather to the fusion rank / all-gather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args, sd, voxd = _frame()
    types = TYPES[:n_agents]
    parts = partition_agents(n_agents, world)
    counts = [len(p) for p in parts]
    mine = parts[rank]
    dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine]) if len(mine) else None
    from airv2x_perception_amd.opencood_iface.sharded import ShardedPipeline
    pipe = ShardedPipeline([OracleBackend(sd, args), OracleBackend(sd, args)], rotate=rotate)
    outs = []
    with torch.no_grad():
        for t in range(world):      # frame t's ego stage runs on rank t % world when rotating
            outs.append(pipe.submit(dd_local, counts=counts)[0])
    pipe.drain()
    if rotate:
        assert [o is not None for o in outs] == [t == rank for t in range(world)]
    else:
        assert all(o is not None for o in outs)
    mineout = outs[rank]
    torch.save({k: mineout[k] for k in ("psm", "rm", "obj", "com", "comm_rate")}, f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_agents,rotate,gather", [(3, 4, True, False), (2, 3, False, False), (3, 2, True, False),
                                                         (3, 4, True, True), (3, 2, True, True)])
def test_uneven_agent_counts_and_rotating_ego_stage(tmp_path, world, n_agents, rotate, gather):
    """4 agents on 3 ranks ([2,1,1]), 3 on 2 ([2,1]) and 2 on 3 ([1,1,0]: an idle rank sends only padding -- NaNs in the
    oracle backend, so a fusion that read them would fail): every rank's frame equals the single-process forward."""
    path = str(tmp_path / "out.pt")
    mp.spawn(_uneven_worker, args=(world, _free_port(), path, n_agents, rotate, gather), nprocs=world, join=True)
    args, sd, voxd = _frame()
    dd = synth.build_data_dict(voxd[:n_agents], TYPES[:n_agents])
    with torch.no_grad():
        ref = orc.where2com_forward(dd, sd, args)
    for r in range(world):
        got = torch.load(f"{path}.{r}")
        for k in ("psm", "rm", "obj"):
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-4), (r, k)
        assert got["comm_rate"] == ref["comm_rate"]
        assert abs(float(got["com"]) - float(ref["com"])) < 1e-6


def test_partition_agents():
    assert [list(r) for r in partition_agents(8, 4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert [list(r) for r in partition_agents(4, 1)] == [[0, 1, 2, 3]]
    assert [list(r) for r in partition_agents(5, 2)] == [[0, 1, 2], [3, 4]]
    assert [list(r) for r in partition_agents(5, 4)] == [[0, 1], [2], [3], [4]]
    assert [list(r) for r in partition_agents(4, 8)] == [[0], [1], [2], [3], [], [], [], []]
    assert [list(r) for r in partition_agents(1, 2)] == [[0], []]
    with pytest.raises(ValueError):
        partition_agents(0, 2)


def test_single_rank_needs_no_process_group():
    args, sd, voxd = _frame()
    dd = synth.build_data_dict(voxd, TYPES)
    with torch.no_grad():
        out = ShardedFrame(OracleBackend(sd, args)).forward(dd)
        ref = orc.where2com_forward(dd, sd, args)
    assert torch.allclose(out["rm"], ref["rm"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("W,world", [(352, 8), (352, 2), (352, 5), (64, 3), (16, 4)])
def test_fusion_column_shards_are_closed_under_both_partitions(W, world):
    """Every rank's column set must contain whole 4-column windows AND whole grid groups {w2 * W/4 + y}: then the window
    and the grid attention of every SwapFusion block stay inside the rank (no exchange), and the sets tile the map."""
    from airv2x_perception_amd.opencood_iface.sharded import fusion_column_shards
    ws, Y = 4, W // 4
    shards = fusion_column_shards(W, ws, world)
    seen = []
    for cols, valid in shards:
        assert len(cols) == len(shards[0][0]) and len(cols) % (ws * ws) == 0          # same shape on every rank
        strip = len(cols) // ws
        real = {c for k, c in enumerate(cols) if k % strip < valid}
        for c in real:
            assert all(ws * (c // ws) + j in real for j in range(ws))                 # its window's columns
            assert all(w2 * Y + c % Y in real for w2 in range(ws))                    # its grid group's columns
        # the compact map keeps both groupings: compact column k = w2' * strip + y' <-> original w2' * Y + y
        for k, c in enumerate(cols):
            assert c // Y == k // strip and (c % Y) % ws == (k % strip) % ws
        seen += sorted(real)
    assert sorted(seen) == list(range(W))
    with pytest.raises(ValueError):
        fusion_column_shards(100, 4, 2)
