"""Agent-sharded collaborative frame: the agents of ONE frame are split over the ranks of a
``torch.distributed`` group (one process per GPU; backend "nccl" = RCCL over xGMI), and the
reference's in-process "communication" — ``Airv2xBase.merge_output_dict_list`` / ``regroup``
(models/common_modules/airv2x_base_model.py:250-283, models/where2comm_modules/where2comm_fuse.py:193-196)
— becomes ONE all-gather of each rank's masked multi-scale feature maps.

The compute is delegated to a *backend* with two methods (``Where2ComEngine`` implements them on
the GPU; the gloo/CPU tests plug in an oracle-based backend to exercise exactly this file):

    local_stage(data_dict_local, has_ego, n_pad=k) -> (send: flat f32 tensor, stats: int64[2], meta)
    ego_stage(recv: flat f32 tensor [world * send.numel()], stats, meta, world) -> output dict

Agent order: the global frame order is [vehicles.., rsus.., drones..] with the ego = vehicle 0
(intermediate_fusion_dataset.py:129-134); rank r owns the contiguous global slice
``partition_agents(n, world)[r]``, so the ego is always local agent 0 of rank 0 and the gathered
buffer is in frame order (rank-major = agent-major).

Uneven frames (5 agents on 4 GPUs, 4 agents on 8): the slices are balanced (sizes differ by at most
one, ranks beyond the agent count own nothing), every rank's message is sized for ``n_pad`` =
the largest local count (an all-gather needs equal contributions) and ``meta["counts"]`` tells the
fusion which slots of the gathered buffer hold agents.  A rank without agents still takes part
in the collectives with an all-padding message.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def partition_agents(n_agents, world):
    """Balanced contiguous slices of the frame order: the first ``n_agents % world`` ranks own one agent more;
    ranks beyond the agent count own an empty range.  The ego (agent 0) is always on rank 0."""
    if n_agents < 1 or world < 1:
        raise ValueError(f"cannot shard {n_agents} agents over {world} ranks")
    q, r = divmod(n_agents, world)
    out, a = [], 0
    for k in range(world):
        c = q + (1 if k < r else 0)
        out.append(range(a, a + c))
        a += c
    return out


def valid_slots(counts, n_pad):
    """Indices (into the rank-major gathered agent axis of world * n_pad slots) of the real agents, in frame order."""
    return [r * n_pad + j for r, c in enumerate(counts) for j in range(c)]


GATHER_MIN_WORLD = 4


class ShardedFrame:
    def __init__(self, backend, group=None, collectives_when_single=False):
        self.backend = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # world == 1 normally skips the collectives; True issues them anyway (a 1-rank RCCL group on a one-GPU box
        # exercises the communicator set-up and the all_gather_into_tensor call path)
        self.collectives_when_single = collectives_when_single and dist.is_initialized()
        # rotating ego stage (fusion_rank given): the messages are GATHERED TO that rank (1/world of the bytes of the all-gather, and the other
        # ranks' links stay free for the next frame's exchange) from GATHER_MIN_WORLD ranks on -- below that the all-gather moves at most twice
        # the bytes and is the collective RCCL is tuned for.  AV2X_SHARD_GATHER=1 / 0 forces either path (the default is unmeasured on
        # hardware: no multi-GPU node has been available to this build; both paths are covered over gloo at world 2 / 3 / 8).
        env = os.environ.get("AV2X_SHARD_GATHER", "auto")
        self.gather_to_fusion_rank = (self.world >= GATHER_MIN_WORLD) if env not in ("0", "1") else env == "1"

    @torch.no_grad()
    def forward(self, data_dict_local, counts=None, fusion_rank=None, **kw):
        """One agent-sharded frame.

        counts       agents per rank (``[len(s) for s in partition_agents(n, world)]``); None = every rank holds as
                     many agents as this one (the even case, no padding).
        fusion_rank  None: every rank finishes the frame (SPMD: all of them return the output dict).  r: only rank r
                     runs the single-level ego stage and returns the output, the others return None right after the
                     exchange (a gather TO rank r: only it needs the messages) — with several frames in flight the caller
                     rotates r so that the ego stages of consecutive frames run on different GPUs instead of being repeated
                     on all of them.  Two-level
                     backends (CoBEVT / V2X-ViT split the fusion itself over the ranks) ignore it."""
        if counts is not None:
            counts = [int(c) for c in counts]
            if len(counts) != self.world:
                raise ValueError(f"counts has {len(counts)} entries for a world of {self.world}")
            n_pad = max(counts)
        else:
            n_pad = None
        if isinstance(data_dict_local, dict):
            data_dict_local.setdefault("shard_rank", self.rank)
            if counts is not None:   # global index of this rank's first agent (When2com's warp matrices)
                data_dict_local["shard_agent_offset"] = sum(counts[:self.rank])
        lkw = {} if n_pad is None else {"n_pad": n_pad}
        send, stats, meta = self.backend.local_stage(data_dict_local, has_ego=(self.rank == 0), **lkw)
        # what this rank puts on the wire for this frame (bench.py prints it next to --dry-run's prediction)
        self.last_exchange = {"message_bytes": int(send.numel() * send.element_size()), "message_dtype": str(send.dtype).replace("torch.", ""),
                              "world": self.world, "backend": (dist.get_backend(self.group) if dist.is_initialized() else None)}
        if counts is not None:
            meta = dict(meta, counts=counts, n_pad=n_pad)
        two_level = (self.world > 1 and getattr(self.backend, "two_level", False)
                     and getattr(self.backend, "can_split", lambda m, w: True)(meta, self.world))
        if self.world == 1 and not self.collectives_when_single:
            recv = send
        elif fusion_rank is not None and not two_level and self.gather_to_fusion_rank:
            # only rank `fusion_rank` finishes this frame, so only IT needs the messages: a gather (point-to-point sends over
            # the peers' own xGMI links into one GPU) moves 1/world of the bytes an all-gather would, and consecutive frames
            # target different ranks.  The counters follow the same route.
            dst = dist.get_global_rank(self.group, fusion_rank) if self.group is not None else fusion_rank
            recv = None
            if self.rank == fusion_rank:
                rb = getattr(self.backend, "recv_buffer", None)
                recv = rb(self.world * send.numel(), send) if rb is not None else \
                    torch.empty(self.world * send.numel(), dtype=send.dtype, device=send.device)
            self.last_exchange["collective"] = "gather to the fusion rank"
            dist.gather(send, list(recv.view(self.world, -1).unbind(0)) if recv is not None else None, dst=dst, group=self.group)
            dist.reduce(stats, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
        else:
            rb = getattr(self.backend, "recv_buffer", None)
            recv = rb(self.world * send.numel(), send) if rb is not None else \
                torch.empty(self.world * send.numel(), dtype=send.dtype, device=send.device)
            # the feature-sharing step: every rank contributes 15.77 MB per agent (default grid);
            # xGMI is point-to-point, so the 7 peer transfers into each GPU run on separate links
            self.last_exchange["collective"] = "all_gather_into_tensor"
            dist.all_gather_into_tensor(recv, send, group=self.group)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        if two_level:
            if hasattr(self.backend, "engine"):
                self.backend.engine.shard_group = self.group
            # second level: every rank runs the fusion on ITS share of the map and the (small) head outputs are
            # gathered, instead of every rank repeating the whole fusion
            part, ctx = self.backend.ego_partial(recv, stats, meta, self.world, self.rank)
            self.last_exchange["second_level_bytes"] = int(part.numel() * part.element_size())
            parts = torch.empty(self.world * part.numel(), dtype=part.dtype, device=part.device)
            dist.all_gather_into_tensor(parts, part, group=self.group)
            return self.backend.ego_finish(parts, ctx, self.world, **kw)
        if fusion_rank is not None and fusion_rank != self.rank:
            return None
        return self.backend.ego_stage(recv, stats, meta, self.world, **kw)


class EngineBackend:
    """Adapter: Where2ComEngine as the ShardedFrame backend."""

    def __init__(self, engine, throughput=None):
        self.engine = engine
        # the mode of the agent-sharded frame, applied around every stage (engine.frame_mode) instead of being left on the caller's engine:
        # sharded_frame = the latency-mode Winograd classes on every rank (one or two agents per rank never fill the chip);
        # throughput = tuner hint only here (bit-identical candidates).  None: the engine's own flag.
        self.throughput = throughput

    def _mode(self):
        e = self.engine
        return e.frame_mode(e.throughput_mode if self.throughput is None else self.throughput, True)

    def local_stage(self, data_dict_local, has_ego, **kw):
        with self._mode():
            return self.engine.shard_local_stage(data_dict_local, has_ego, **kw)

    def ego_stage(self, recv, stats, meta, world, **kw):
        with self._mode():
            return self.engine.shard_ego_stage(recv, stats, meta, world, **kw)

    def recv_buffer(self, numel, like):
        """The all-gather destination out of the engine's workspace pool (no allocator traffic per frame)."""
        return self.engine.buf("shard_recv", (numel,), like.dtype)

    @property
    def two_level(self):
        return hasattr(self.engine, "shard_ego_partial") and getattr(self.engine, "fusion_sharding", True)

    def can_split(self, meta, world):
        """V2X-ViT needs the map to split into equal strips of whole windows; CoBEVT pads, so it always can."""
        fs = getattr(self.engine, "fusion_strip", None)
        return fs is None or fs(meta["W"], world, 0) is not None

    def ego_partial(self, recv, stats, meta, world, rank):
        with self._mode():
            return self.engine.shard_ego_partial(recv, stats, meta, world, rank)

    def ego_finish(self, parts, ctx, world, **kw):
        with self._mode():
            return self.engine.shard_ego_finish(parts, ctx, world, **kw)


class ShardedPipeline:
    """``depth`` agent-sharded frames in flight per rank: frame t's all-gather (RCCL's own stream) and ego stage overlap
    the local stage of frame t+1, which runs on another HIP stream with its own workspaces (weights shared).  Every rank
    must submit the same frames in the same order (collectives are matched by issue order).

    ``rotate``: the single-level ego stage of frame t runs on rank t % world only (ShardedFrame's ``fusion_rank``), so a
    group of N GPUs finishes N frames' fusions concurrently instead of repeating each one N times; the output of frame t
    is returned by that rank's ``submit`` and is None elsewhere.

    ``backends``: one backend per in-flight slot (the GPU engines come from ``engine.share_weights()``; CPU / gloo tests
    pass oracle backends, for which no streams are used)."""

    def __init__(self, backends, group=None, rotate=True, device=None):
        self.frames = [ShardedFrame(b, group) for b in backends]
        self.world, self.rank = self.frames[0].world, self.frames[0].rank
        self.rotate = rotate
        self.cuda = device is not None and torch.device(device).type == "cuda"
        if self.cuda:      # from the process-wide pool (engine.pooled_streams): HIP has four hardware queues, a stream too many costs 25 %
            from .engine import pooled_streams
            self.streams = pooled_streams(device, len(backends))
        else:
            self.streams = [None] * len(backends)
        self.t = 0

    @classmethod
    def from_engine(cls, engine, depth, group=None, rotate=True):
        # depth > 1: tuner hint (bit-identical candidates); sharded_frame: one or two agents per rank -> the latency-mode Winograd classes
        # (engine.wino4_rule), every rank alike.  Both live on the backends and are applied around each stage: the caller's engine keeps its flags.
        engines = [engine] + [engine.share_weights() for _ in range(depth - 1)]
        return cls([EngineBackend(e, throughput=depth > 1) for e in engines], group, rotate, engine.device)

    def submit(self, data_dict_local, counts=None, **kw):
        """Enqueue one frame; returns (output dict or None, event on the frame's stream or None)."""
        k = self.t % len(self.frames)
        fr = (self.t % self.world) if (self.rotate and self.world > 1) else None
        self.t += 1
        s = self.streams[k]
        if s is None:
            return self.frames[k].forward(data_dict_local, counts=counts, fusion_rank=fr, **kw), None
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out = self.frames[k].forward(data_dict_local, counts=counts, fusion_rank=fr, **kw)
            ev = torch.cuda.Event()
            ev.record(s)
        return out, ev

    def drain(self):
        for s in self.streams:
            if s is not None:
                torch.cuda.current_stream().wait_stream(s)


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
