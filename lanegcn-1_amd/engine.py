"""Lean forward engine of the hot path: flat, pre-collated device batch in, stage features out.

``collate_flat`` does on the host, once per batch, what the reference spreads over ~1.3 k tiny
H2D copies and casts (utils.gpu / utils.to_long, lanegcn.py:129-134): every per-scene array is
concatenated into one flat buffer and uploaded with one copy per array.  ``HotPathEngine.forward``
then runs graph_gather -> CSR plan -> MapNet -> A2M -> M2M -> M2A -> A2A with no host
synchronisation and no data-dependent host control flow, so the whole forward can be captured in
a hipGraph (``capture``) and replayed with one host call.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import os

import numpy as np
import torch

from . import ops
from . import lanegcn as M


@dataclass
class FlatBatch:
    """One batch of scenes, flat and device resident."""
    n_scenes: int
    n_nodes: int
    n_actors: int
    num_scales: int
    node_ctrs: torch.Tensor     # [N,2]
    node_feats: torch.Tensor    # [N,2]
    turn: torch.Tensor          # [N,2]
    control: torch.Tensor       # [N]
    intersect: torch.Tensor     # [N]
    actor_ctrs: torch.Tensor    # [A,2]
    node_off: torch.Tensor      # [B+1] int32
    actor_off: torch.Tensor     # [B+1] int32
    idx_local: torch.Tensor     # [2*sumE] int64: scene-local u/v of every (relation, u|v, scene) segment
    seg_off: torch.Tensor       # [S+1] int64
    seg_base: torch.Tensor      # [S] int64: node offset of the segment's scene
    rel_slices: List[tuple]     # per relation: ((u_begin, u_end), (v_begin, v_end)) into idx_local
    cap_a2m: int                # sum_i n_i * a_i   (upper bound of the pair counts)
    cap_a2a: int                # sum_i a_i * a_i
    n_edges: List[int]
    _buf: Optional[torch.Tensor] = None     # the one device byte buffer the arrays are views of (collate_flat)

    def buf_view(self) -> Optional[torch.Tensor]:
        return self._buf


_FLAT_ARRAYS = ("node_ctrs", "node_feats", "turn", "control", "intersect", "actor_ctrs", "node_off", "actor_off",
                "idx_local", "seg_off", "seg_base")


@dataclass
class HostFlatBatch:
    """One batch of scenes, flat, on the HOST: every array of FlatBatch packed at 256-byte aligned offsets into ONE
    byte buffer (pinned when a GPU is present), so that uploading a batch is one host-to-device copy
    (the reference does ~1.3 k per batch, utils.py:74-96).  Built by ``collate_flat_host`` -- plain
    numpy, so a DataLoader worker can do it (``collate_fn`` of this package attaches it as data["_flat"])."""
    buf: torch.Tensor                      # uint8 [nbytes]
    layout: Dict[str, tuple]               # name -> (offset, numpy dtype str, shape)
    meta: Dict                             # the non-array fields of FlatBatch

    def to(self, device=None) -> "FlatBatch":
        """Upload (one copy) and cut the device buffer into FlatBatch views (256-byte aligned)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        dbuf = self.buf if dev.type == "cpu" else self.buf.to(dev, non_blocking=True)
        arrs = {}
        for name, (off, dt, shape) in self.layout.items():
            cnt = int(np.prod(shape)) if len(shape) else 1
            tdt = {"float32": torch.float32, "int32": torch.int32, "int64": torch.int64}[dt]
            arrs[name] = dbuf[off:off + cnt * np.dtype(dt).itemsize].view(tdt).view(*shape)
        return FlatBatch(**self.meta, **arrs, _buf=dbuf)


def collate_flat_host(scenes: List[Dict], pin: Optional[bool] = None) -> HostFlatBatch:
    """Host collate of scene dicts (numpy or CPU torch leaves) into one staging buffer."""

    def npy(x):
        return x.numpy() if torch.is_tensor(x) else np.asarray(x)

    graphs = [s["graph"] for s in scenes]
    B = len(scenes)
    ns = len(graphs[0]["pre"])
    n_nodes = np.array([int(g["num_nodes"]) for g in graphs], np.int64)
    n_act = np.array([len(s["ctrs"]) for s in scenes], np.int64)
    node_off = np.zeros(B + 1, np.int64)
    np.cumsum(n_nodes, out=node_off[1:])
    actor_off = np.zeros(B + 1, np.int64)
    np.cumsum(n_act, out=actor_off[1:])

    pieces, seg_len, seg_base, rel_slices, n_edges = [], [], [], [], []
    pos = 0

    def add(getter):
        nonlocal pos
        begin = pos
        for j, g in enumerate(graphs):
            x = npy(getter(g)).astype(np.int64, copy=False).reshape(-1)   # 0-dim guard (lanegcn.py:203-207) + to_long
            pieces.append(x)
            seg_len.append(len(x))
            seg_base.append(node_off[j])
            pos += len(x)
        return (begin, pos)

    keys = []
    for i in range(ns):
        keys += [("pre", i), ("suc", i)]
    for k1, i in keys:
        su = add(lambda g: g[k1][i]["u"])
        sv = add(lambda g: g[k1][i]["v"])
        rel_slices.append((su, sv))
        n_edges.append(su[1] - su[0])
    for k1 in ("left", "right"):
        su = add(lambda g: g[k1]["u"])
        sv = add(lambda g: g[k1]["v"])
        rel_slices.append((su, sv))
        n_edges.append(su[1] - su[0])
    seg_off = np.zeros(len(seg_len) + 1, np.int64)
    np.cumsum(seg_len, out=seg_off[1:])

    cat = lambda key, src: np.concatenate([npy(s[key]) for s in src], 0)
    arrays = {
        "node_ctrs": cat("ctrs", graphs).astype(np.float32, copy=False), "node_feats": cat("feats", graphs).astype(np.float32, copy=False),
        "turn": cat("turn", graphs).astype(np.float32, copy=False), "control": cat("control", graphs).astype(np.float32, copy=False),
        "intersect": cat("intersect", graphs).astype(np.float32, copy=False), "actor_ctrs": cat("ctrs", scenes).astype(np.float32, copy=False),
        "node_off": node_off.astype(np.int32), "actor_off": actor_off.astype(np.int32),
        "idx_local": np.concatenate(pieces) if pieces else np.zeros(0, np.int64),
        "seg_off": seg_off, "seg_base": np.asarray(seg_base, np.int64),
    }
    layout, off = {}, 0
    for name in _FLAT_ARRAYS:
        a = arrays[name]
        layout[name] = (off, str(a.dtype), tuple(a.shape))
        off = (off + a.nbytes + 255) & ~255
    pin = torch.cuda.is_available() if pin is None else pin
    buf = torch.empty(max(off, 256), dtype=torch.uint8, pin_memory=bool(pin))
    view = buf.numpy()
    for name in _FLAT_ARRAYS:
        o, _, _ = layout[name]
        a = np.ascontiguousarray(arrays[name])
        view[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    meta = dict(n_scenes=B, n_nodes=int(node_off[-1]), n_actors=int(actor_off[-1]), num_scales=ns,
                rel_slices=rel_slices, cap_a2m=int(np.dot(n_nodes, n_act)), cap_a2a=int(np.dot(n_act, n_act)),
                n_edges=n_edges)
    return HostFlatBatch(buf, layout, meta)


def collate_flat(scenes: List[Dict], device=None, pin: Optional[bool] = None) -> FlatBatch:
    """Host collate of scene dicts into a FlatBatch on `device`: one staging buffer, ONE host-to-device copy.
    (Round 1 reported that packing the arrays as views of staging buffers halved the throughput of four captured
    forwards in flight; with every view 256-byte aligned in one buffer that does not reproduce: DESIGN.md 5c.)"""
    on_cpu = device is not None and torch.device(device).type == "cpu"
    return collate_flat_host(scenes, pin=False if on_cpu else pin).to(device)


def host_actor_inputs(scenes):
    """[A,3,20] actor tracks (actor_gather, lanegcn.py:155-168) and the per-actor world-frame transform rot [A,2,2] /
    orig [A,2] (the scene's, repeated for its actors) as host tensors + the actors-per-scene list."""
    npy = lambda x: x.numpy() if torch.is_tensor(x) else np.asarray(x)
    feats = np.concatenate([npy(s["feats"]).transpose(0, 2, 1) for s in scenes], 0).astype(np.float32)
    rot = np.concatenate([np.repeat(npy(s["rot"])[None], len(s["ctrs"]), 0) for s in scenes]).astype(np.float32)
    orig = np.concatenate([np.repeat(npy(s["orig"])[None], len(s["ctrs"]), 0) for s in scenes]).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return t(feats), t(rot), t(orig), [len(s["ctrs"]) for s in scenes]


class HotPathEngine:
    """graph_gather -> MapNet -> A2M -> M2M -> M2A -> A2A on the HIP kernels, sync-free."""

    def __init__(self, map_net: M.MapNet, a2m: M.A2M, m2m: M.M2M, m2a: M.M2A, a2a: M.A2A, config=None,
                 legacy_offsets: bool = True, branches: bool = False, lane_impl: Optional[str] = None):
        self.map_net, self.a2m, self.m2m, self.m2a, self.a2a = map_net, a2m, m2m, m2a, a2a
        self.config = config or M.config
        self.legacy_offsets = legacy_offsets
        # LaneConv implementation (lanegcn.lane_conv): None = the package default ("tiled": weight-stationary row
        # blocks, the faster one for ONE forward at a time); "fused" = one elastic launch per layer, which packs
        # better when several captured forwards share the GPU (bench.py --streams > 1)
        self.lane_impl = lane_impl
        # optional parallel graph branches: the three pair searches (index work that depends only on the centres)
        # and every Att's V GEMM on a side stream, forked / joined with events (captured as graph edges).
        # Measured on MI355X / ROCm 7.2: bitwise the same result, no single-stream gain (38.9 k vs 39.4 k
        # scenes/s) and a loss with four graphs in flight (61 k vs 90 k) -> off by default.
        self.branches = branches
        self._side = None
        # the integer stage as lgcn_index_build (4 launches) instead of 12; False: the separate entry points
        self.fused_index = os.environ.get("LGCN_INDEX", "fused") == "fused"
        self._cnt_capture = None      # counter buffer of the forward being captured (capture(); it lives with the graph)
        # Pair capacities (rows of hi / wi and of every [cap, 128] pair-row buffer of a forward).  "bound": sum_i t_i s_i,
        # can never overflow (265 MB of pair rows per A2M layer at S2, untouched but reserved).  "tight" (default):
        # 1.25 x the largest pair count this engine has seen per set, at least 4,096 (the bound until a count is known)
        # -- learnt by capture() from its eager warm-up forwards (one host read each) and by forward_guarded() /
        # Net.forward's host read; a forward whose count exceeds it comes back with a negative n_pairs (the kernels keep
        # to the capacity), learn_pair_counts(out) says so and grows the capacity, the caller runs again; a captured
        # graph has to be captured again.
        self.pair_caps = os.environ.get("LGCN_PAIR_CAPS", "tight")
        self._pair_seen = [0, 0, 0]

    def _counters(self, fb: FlatBatch) -> torch.Tensor:
        """Key counters of lgcn_index_build (zero before, zero after).  A captured forward owns a buffer of its own
        (capture() sets it); eager forwards share one per FlatBatch -- do not run two eager forwards of the SAME
        FlatBatch concurrently on different streams."""
        if self._cnt_capture is not None:
            return self._cnt_capture
        cnt = fb.__dict__.get("_idx_cnt")
        if cnt is None:
            cnt = ops.index_counters(fb.n_nodes, len(fb.rel_slices), fb.node_ctrs.device)
            fb.__dict__["_idx_cnt"] = cnt
        return cnt

    @torch.no_grad()
    def forward(self, fb: FlatBatch, actors: torch.Tensor, stages: bool = False,
                mapnet_only: bool = False) -> Dict[str, torch.Tensor]:
        """actors: [A,128] ActorNet output.  Returns {"nodes", "actors"} (+ every stage if stages).
        mapnet_only: stop after MapNet (BASELINE config "MapNet LaneConv only"): graph_gather + plan + MapNet."""
        if mapnet_only:
            g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
            plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices],
                                 fb.n_nodes)
            feat = M.lane_conv(self.map_net.fuse, self.map_net.stem(fb.node_ctrs, fb.node_feats), plan, fb.num_scales,
                               impl=self.lane_impl)
            return {"nodes": feat, "actors": actors}
        cfg = self.config
        out = {}
        main = torch.cuda.current_stream()
        side = None
        if self.branches:
            if self._side is None:
                self._side = torch.cuda.Stream()
            side = self._side
        dev = fb.node_ctrs.device
        searches = ((fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m),
                    (fb.actor_ctrs, fb.actor_off, fb.node_ctrs, fb.node_off, cfg["map2actor_dist"], fb.cap_a2m),
                    (fb.actor_ctrs, fb.actor_off, fb.actor_ctrs, fb.actor_off, cfg["actor2actor_dist"], fb.cap_a2a))
        if self.pair_caps == "tight":      # nothing seen yet for a set: the bound (the first forward cannot overflow)
            searches = tuple(s[:5] + (min(s[5], self._cap(i)) if self._pair_seen[i] > 0 else s[5],) for i, s in enumerate(searches))
        bufs = [ops.pairs_alloc(s[0].shape[0], fb.n_scenes, s[5], dev) for s in searches]   # on the main stream
        if side is not None:
            side.wait_stream(main)
        if side is None and self.fused_index and ops.index_fused_ok(fb.n_nodes, len(fb.rel_slices), sum(fb.n_edges)):
            # graph_gather (lanegcn.py:171-209) + CSR plan + the three pair searches: four launches in all
            flag = torch.empty(1, dtype=torch.int32, device=dev)       # the range guard's flag, cleared by the first launch
            plan, pairs = ops.index_build(fb.idx_local, fb.seg_off, fb.seg_base, fb.rel_slices, fb.n_nodes, searches,
                                          self.legacy_offsets, bufs=bufs, cnt=self._counters(fb), clear_word=flag)
        else:
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.stream(side if side is not None else main):
                pairs = ops.pairs_build_multi(searches, self.legacy_offsets, bufs=bufs)   # three sets, three launches
            g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
            us = [g64[a:b] for (a, b), _ in fb.rel_slices]
            vs = [g64[a:b] for _, (a, b) in fb.rel_slices]
            plan = ops.csr_build(us, vs, fb.n_nodes)
        # MapNet (lanegcn.py:311-363)
        feat = self.map_net.stem(fb.node_ctrs, fb.node_feats)
        feat = M.lane_conv(self.map_net.fuse, feat, plan, fb.num_scales, impl=self.lane_impl)
        if stages:
            out["map_net"] = feat
        # A2M (lanegcn.py:385-407)
        fold = side is None and M._fold_ok(actors) and fb.n_nodes > 0
        if fold:      # launches folded across the Att layers of the three blocks (lanegcn.att_block): 8 row-block launches
            feat = self.a2m.run(feat, fb.turn, fb.control, fb.intersect, actors, pairs[0])
        else:
            feat = self.a2m.fuse_meta(feat, fb.turn, fb.control, fb.intersect)
            if side is not None:
                main.wait_stream(side)            # the pair sets are needed from here on
            for att in self.a2m.att:
                feat = att.run(feat, actors, pairs[0], side)
        if stages:
            out["a2m"] = feat
        # M2M (lanegcn.py:445-480)
        feat = M.lane_conv(self.m2m.fuse, feat, plan, fb.num_scales, impl=self.lane_impl)
        if stages:
            out["m2m"] = feat
        # M2A (lanegcn.py:502-513)
        act = actors
        if fold:
            # M2A's last tail also emits U / V of A2A's first layer (its targets and context are M2A's output rows)
            act, uv = M.att_block(self.m2a.att, act, feat, pairs[1], next_att=self.a2a.att[0], next_ctx_is_out=True)
            if stages:
                out["m2a"] = act
            act, _ = M.att_block(self.a2a.att, act, act, pairs[2], uv=uv, ctx_is_agts=True)
        else:
            for att in self.m2a.att:
                act = att.run(act, feat, pairs[1], side)
            if stages:
                out["m2a"] = act
            # A2A (lanegcn.py:534-545)
            for att in self.a2a.att:
                act = att.run(act, act, pairs[2], side)
        if stages:
            out["a2a"] = act
        out["nodes"], out["actors"] = feat, act
        out["n_pairs"] = [p.n_pairs for p in pairs]
        # range check of the 16-bit-plane modes, on the device and part of every (captured) forward: bit 0 of
        # out["nonfinite"] is set when a feature row came out NaN / inf (ops.guarded / forward_guarded act on it)
        ops.check_finite(flag, feat, act)
        out["nonfinite"] = flag
        return out

    def _cap(self, i: int) -> int:
        """Tight capacity of pair set i (A2M, M2A, A2A): 1.25 x the largest count seen, at least 4,096, in 1,024s."""
        want = max(4096, int(self._pair_seen[i] * 1.25) + 1)
        return (want + 1023) // 1024 * 1024

    def learn_pair_counts(self, out: Dict[str, torch.Tensor]) -> bool:
        """Host side of the tight pair capacities: reads the forward's three pair counts (one device->host copy), keeps
        the largest per set; True when a count exceeded its capacity (negative n_pairs: the forward's features are
        those of a truncated pair set and must be recomputed with the grown capacity)."""
        counts = torch.stack([c.view(()) for c in out["n_pairs"]]).tolist()
        over = False
        for i, c in enumerate(counts):
            over = over or c < 0
            self._pair_seen[i] = max(self._pair_seen[i], abs(int(c)))
        return over

    def forward_guarded(self, fb: FlatBatch, actors: torch.Tensor, **kw) -> Dict[str, torch.Tensor]:
        """forward() + the host side of the range guard: reads the flag (one 4-byte device->host copy) and re-runs an
        f16x2 forward that left fp16's range in bf16x3, or raises, as ops.set_guard() says.  With tight pair capacities
        it also reads the pair counts and re-runs a forward that overflowed them."""
        out = self.forward(fb, actors, **kw)
        if self.pair_caps == "tight" and not kw.get("mapnet_only") and self.learn_pair_counts(out):
            out = self.forward(fb, actors, **kw)
            if self.learn_pair_counts(out):
                raise ops.L.LgcnError("pair capacity still exceeded after growing it")
        if ops.get_guard() == "off" or ops.get_mma() != "f16x2" or int(out["nonfinite"].item()) == 0:
            return out
        if ops.get_guard() == "raise":
            raise ops.L.LgcnError("non-finite features in f16x2 mode: an operand left fp16's range (|x| >= 65504)")
        with ops.mma_scope("bf16x3"):
            return self.forward(fb, actors, **kw)

    @torch.no_grad()
    def stage_functions(self, fb: FlatBatch, actors: torch.Tensor):
        """The same forward cut at the reference's module boundaries, for per-stage measurement: returns
        (state, [(name, fn)]); fn() runs one stage on the tensors the previous stages left in `state`
        (run them in order once before capturing any of them in a graph)."""
        cfg, st = self.config, {}

        def index():
            searches = ((fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m),
                        (fb.actor_ctrs, fb.actor_off, fb.node_ctrs, fb.node_off, cfg["map2actor_dist"], fb.cap_a2m),
                        (fb.actor_ctrs, fb.actor_off, fb.actor_ctrs, fb.actor_off, cfg["actor2actor_dist"], fb.cap_a2a))
            if self.fused_index and ops.index_fused_ok(fb.n_nodes, len(fb.rel_slices), sum(fb.n_edges)):
                cnt = st.get("_cnt")             # the stage's own counters (its captured graph is replayed alone)
                if cnt is None:
                    cnt = st["_cnt"] = ops.index_counters(fb.n_nodes, len(fb.rel_slices), fb.node_ctrs.device)
                st["plan"], st["pairs"] = ops.index_build(fb.idx_local, fb.seg_off, fb.seg_base, fb.rel_slices, fb.n_nodes,
                                                          searches, self.legacy_offsets, cnt=cnt)
                return
            st["pairs"] = ops.pairs_build_multi(searches, self.legacy_offsets)
            g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
            st["plan"] = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices],
                                       [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)

        def map_net():
            st["nodes"] = M.lane_conv(self.map_net.fuse, self.map_net.stem(fb.node_ctrs, fb.node_feats), st["plan"],
                                      fb.num_scales, impl=self.lane_impl)

        def a2m():
            st["nodes_a2m"] = self.a2m.run(st["nodes"], fb.turn, fb.control, fb.intersect, actors, st["pairs"][0])

        def m2m():
            st["nodes_m2m"] = M.lane_conv(self.m2m.fuse, st["nodes_a2m"], st["plan"], fb.num_scales, impl=self.lane_impl)

        def m2a():      # as in forward(): the last tail also emits A2A's first U / V
            if M._fold_ok(st["nodes_m2m"]):
                st["actors_m2a"], st["uv_a2a"] = M.att_block(self.m2a.att, actors, st["nodes_m2m"], st["pairs"][1],
                                                             next_att=self.a2a.att[0], next_ctx_is_out=True)
                return
            act = actors
            for att in self.m2a.att:
                act = att.run(act, st["nodes_m2m"], st["pairs"][1])
            st["actors_m2a"], st["uv_a2a"] = act, None

        def a2a():
            act = st["actors_m2a"]
            if M._fold_ok(act):
                st["actors_a2a"] = M.att_block(self.a2a.att, act, act, st["pairs"][2], uv=st["uv_a2a"], ctx_is_agts=True)[0]
                return
            for att in self.a2a.att:
                act = att.run(act, act, st["pairs"][2])
            st["actors_a2a"] = act

        return st, [("index", index), ("map_net", map_net), ("a2m", a2m), ("m2m", m2m), ("m2a", m2a), ("a2a", a2a)]

    def capture(self, fb: FlatBatch, actors: torch.Tensor, warmup: int = 2, **fwd_kw):
        """Capture one forward into a hipGraph.  Returns (graph, outputs); ``graph.replay()`` re-runs
        the whole forward on the captured buffers (refill fb's tensors / `actors` in place first)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2 if self.pair_caps == "tight" else 0)):
                o = self.forward(fb, actors, **fwd_kw)   # also fills the weight-pack caches outside the capture
                if self.pair_caps == "tight" and not fwd_kw.get("mapnet_only"):
                    self.learn_pair_counts(o)           # the captured buffers are sized for THIS batch's counts x 1.25
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with self.own_counters(fb) as cnt:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(fb, actors, **fwd_kw)
        # a graph holds ADDRESSES: the captured inputs (and its counter buffer) must outlive it -- a freed FlatBatch's
        # memory is handed to the next allocation, and the replay then builds its plan from whatever lies there
        graph._lgcn_inputs = (fb, actors, cnt)
        return graph, out

    def own_counters(self, fb: FlatBatch):
        """Context: forwards inside it use a fresh, zeroed counter buffer (yielded: keep it alive as long as what was
        captured) -- for a forward that is being captured: several captured forwards may replay concurrently."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            cnt = ops.index_counters(fb.n_nodes, len(fb.rel_slices), fb.node_ctrs.device)
            torch.cuda.synchronize()         # zeroed before anything captured can run
            prev, self._cnt_capture = self._cnt_capture, cnt
            try:
                yield cnt
            finally:
                self._cnt_capture = prev
        return cm()


class FullNetEngine:
    """Whole Net.forward (lanegcn.py:127-151) on flat device inputs, capturable in one hipGraph: ActorNet (stock
    conv ops) -> hot path (HotPathEngine) -> PredNet (HIP row blocks + stock heads) -> world-frame transform."""

    def __init__(self, net: "M.Net"):
        self.net = net
        self.hot = HotPathEngine(net.map_net, net.a2m, net.m2m, net.m2a, net.a2a, net.config)

    @staticmethod
    def actor_inputs(scenes, device=None):
        """Device copies of host_actor_inputs(scenes): (feats [A,3,20], rot [A,2,2], orig [A,2])."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        feats, rot, orig, _ = host_actor_inputs(scenes)
        return feats.to(dev), rot.to(dev), orig.to(dev)

    @torch.no_grad()
    def forward(self, fb: FlatBatch, actor_feats: torch.Tensor, rot: torch.Tensor, orig: torch.Tensor,
                sizes: List[int], return_pairs: bool = False) -> Dict[str, torch.Tensor]:
        """Returns {"cls": [A,6], "reg": [A,6,30,2]} for all actors of the batch (scene i = rows
        sum(sizes[:i]) .. sum(sizes[:i+1])), reg already in world coordinates."""
        net = self.net
        actors = net.actor_net(actor_feats)
        hot = self.hot.forward(fb, actors)
        actors = hot["actors"]
        if net.pred_net._hip_ok(actors):
            cls, reg = net.pred_net.forward_flat(actors, fb.actor_ctrs, rot, orig)     # world frame inside the last launch
        else:
            idcs, ctrs, st = [], [], 0
            for n in sizes:
                idcs.append(slice(st, st + n))
                ctrs.append(fb.actor_ctrs[st:st + n])
                st += n
            out = net.pred_net(actors, idcs, ctrs)
            reg = torch.cat(out["reg"], 0)
            cls = torch.cat(out["cls"], 0)
            reg = torch.einsum("amtk,akj->amtj", reg, rot) + orig.view(-1, 1, 1, 2)
        res = {"cls": cls, "reg": reg, "nonfinite": hot["nonfinite"]}
        ops.check_finite(hot["nonfinite"], reg.reshape(-1), cls.reshape(-1), bit=2)      # PredNet's row blocks too
        # device counts of the three pair sets (A2M, M2A, A2A): Att.strict's check and the tight pair capacities
        # (negative = capacity exceeded, HotPathEngine.learn_pair_counts); `return_pairs` is kept for callers of round 2
        res["n_pairs"] = hot["n_pairs"]
        return res

    def capture(self, fb, actor_feats, rot, orig, sizes, warmup: int = 3, tune_convs: bool = True,
                capture_error_mode: str = "global", **fwd_kw):
        """Capture the whole Net forward.  tune_convs: let MIOpen search its solvers for ActorNet's 17 Conv1d shapes
        during the warm-up (torch.backends.cudnn.benchmark): the shapes of a captured graph are fixed, and the
        default heuristic picks were measured 11 % slower end to end (2.34 vs 2.10 ms per batch)."""
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = bool(tune_convs)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    o = self.forward(fb, actor_feats, rot, orig, sizes, **fwd_kw)
                    if self.hot.pair_caps == "tight" and self.hot.learn_pair_counts(o):      # sized for this batch x 1.25
                        self.forward(fb, actor_feats, rot, orig, sizes, **fwd_kw)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with self.hot.own_counters(fb) as cnt:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=capture_error_mode):
                    out = self.forward(fb, actor_feats, rot, orig, sizes, **fwd_kw)
            graph._lgcn_inputs = (fb, actor_feats, rot, orig, cnt)      # the graph holds their addresses: keep them alive
        finally:
            torch.backends.cudnn.benchmark = prev
        return graph, out
