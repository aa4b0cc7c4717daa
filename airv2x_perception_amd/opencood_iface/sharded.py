"""Agent-sharded collaborative frame: the agents of ONE frame are split over the ranks of a
``torch.distributed`` group (one process per GPU; backend "nccl" = RCCL over xGMI), and the
reference's in-process "communication" — ``Airv2xBase.merge_output_dict_list`` / ``regroup``
(models/common_modules/airv2x_base_model.py:250-283, models/where2comm_modules/where2comm_fuse.py:193-196)
— becomes ONE all-gather of each rank's masked multi-scale feature maps.

The compute is delegated to a *backend* with two methods (``Where2ComEngine`` implements them on
the GPU; the gloo/CPU tests plug in an oracle-based backend to exercise exactly this file):

    local_stage(data_dict_local, has_ego) -> (send: flat f32 tensor, stats: int64[2], meta)
    ego_stage(recv: flat f32 tensor [world * send.numel()], stats, meta, world) -> output dict

Agent order: the global frame order is [vehicles.., rsus.., drones..] with the ego = vehicle 0
(intermediate_fusion_dataset.py:129-134); rank r owns the contiguous global slice
``partition_agents(n, world)[r]``, so the ego is always local agent 0 of rank 0 and the gathered
buffer is already in frame order (rank-major = agent-major).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def partition_agents(n_agents, world):
    """Contiguous equal slices (all_gather needs equal counts): n_agents % world must be 0."""
    if n_agents % world != 0 or n_agents < world:
        raise ValueError(f"{n_agents} agents cannot be sharded evenly over {world} ranks")
    k = n_agents // world
    return [range(r * k, (r + 1) * k) for r in range(world)]


class ShardedFrame:
    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    @torch.no_grad()
    def forward(self, data_dict_local, **kw):
        if isinstance(data_dict_local, dict):
            data_dict_local.setdefault("shard_rank", self.rank)   # global agent index = rank * n_loc + j (When2com's warp)
        send, stats, meta = self.backend.local_stage(data_dict_local, has_ego=(self.rank == 0))
        if self.world == 1:
            recv = send
        else:
            recv = torch.empty(self.world * send.numel(), dtype=send.dtype, device=send.device)
            # the feature-sharing step: every rank contributes 15.77 MB per agent (default grid);
            # xGMI is point-to-point, so the 7 peer transfers into each GPU run on separate links
            dist.all_gather_into_tensor(recv, send, group=self.group)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        if (self.world > 1 and getattr(self.backend, "two_level", False)
                and getattr(self.backend, "can_split", lambda m, w: True)(meta, self.world)):
            if hasattr(self.backend, "engine"):
                self.backend.engine.shard_group = self.group
            # second level: every rank runs the fusion on ITS share of the map and the (small) head outputs are
            # gathered, instead of every rank repeating the whole fusion
            part, ctx = self.backend.ego_partial(recv, stats, meta, self.world, self.rank)
            parts = torch.empty(self.world * part.numel(), dtype=part.dtype, device=part.device)
            dist.all_gather_into_tensor(parts, part, group=self.group)
            return self.backend.ego_finish(parts, ctx, self.world, **kw)
        return self.backend.ego_stage(recv, stats, meta, self.world, **kw)


class EngineBackend:
    """Adapter: Where2ComEngine as the ShardedFrame backend."""

    def __init__(self, engine):
        self.engine = engine

    def local_stage(self, data_dict_local, has_ego):
        return self.engine.shard_local_stage(data_dict_local, has_ego)

    def ego_stage(self, recv, stats, meta, world, **kw):
        return self.engine.shard_ego_stage(recv, stats, meta, world, **kw)

    @property
    def two_level(self):
        return hasattr(self.engine, "shard_ego_partial") and getattr(self.engine, "fusion_sharding", True)

    def can_split(self, meta, world):
        """V2X-ViT needs the map to split into equal strips of whole windows; CoBEVT pads, so it always can."""
        fs = getattr(self.engine, "fusion_strip", None)
        return fs is None or fs(meta["W"], world, 0) is not None

    def ego_partial(self, recv, stats, meta, world, rank):
        return self.engine.shard_ego_partial(recv, stats, meta, world, rank)

    def ego_finish(self, parts, ctx, world, **kw):
        return self.engine.shard_ego_finish(parts, ctx, world, **kw)


def fusion_column_shards(W, window, world):
    """Second-level sharding of a fused-axial-attention map (swap_fusion_modules.py:154-195) WITHOUT any exchange between
    the window and the grid halves: with Y = W / window, the window partition groups columns [window*b, window*b + window)
    and the grid partition groups columns {w2 * Y + y : w2 < window}.  When window | Y, the set of columns whose residue
    mod Y falls into the aligned group [window*g, window*g + window) is closed under BOTH groupings (all rows), so the
    G = Y / window residue groups are independent sub-problems.  Rank r takes a contiguous run of ceil(G / world)
    groups (padded by repeating the last group so that every rank solves the same shape; ``valid`` counts the real ones).
    The compacted map (H, window * window * per) keeps both partitions intact, so the ordinary kernels run on it.
    Returns [(column indices into the full map, number of valid compact columns per strip)] per rank."""
    if W % (window * window):
        raise ValueError(f"map width {W} is not a multiple of window^2 = {window * window}: no exchange-free column sharding")
    Y = W // window
    G = Y // window
    per = -(-G // world)
    out = []
    for r in range(world):
        gs = list(range(r * per, min((r + 1) * per, G)))
        valid = len(gs)
        gs = gs + [G - 1] * (per - valid)
        cols = [w2 * Y + window * g + j for w2 in range(window) for g in gs for j in range(window)]
        out.append((cols, valid * window))
    return out
