"""Input side of the hot path: the reference's batch schema (data.py:203-216, 220-361, 555-575), a
from-scratch ``dilated_nbrs`` (data.py:520-534) and the synthetic Argoverse-shaped scene generator
of SURVEY.md section 8(d).  The Argoverse dataset readers themselves are out of scope (no
argoverse-api / dataset in the image)."""
import numpy as np
import torch

NUM_SCALES = 6


def dilated_nbrs(nbr, num_nodes, num_scales):
    """Edges of A^(2^i), i = 1..num_scales-1, of the boolean adjacency A given as nbr = {u, v}.

    Same edge SETS as the reference's scipy ``mat = mat * mat`` chain (data.py:520-534); returned
    sorted by (u, v) (scipy's order inside a row is unspecified)."""
    n = int(num_nodes)
    key = np.unique(np.asarray(nbr["u"], np.int64) * n + np.asarray(nbr["v"], np.int64))
    out = []
    for _ in range(1, num_scales):
        u, m = key // n, key % n
        # rows of A start at start[m]; (u, m) joins every (m, v)
        start = np.searchsorted(key, np.arange(n + 1, dtype=np.int64) * n)
        deg = start[m + 1] - start[m]
        tot = int(deg.sum())
        if tot:
            src = np.repeat(np.arange(len(key)), deg)
            off = np.arange(tot) - np.repeat(np.cumsum(deg) - deg, deg)
            v2 = key[start[m[src]] + off] % n
            key = np.unique(u[src] * n + v2)
        else:
            key = np.zeros(0, np.int64)
        out.append({"u": (key // n).astype(np.int64), "v": (key % n).astype(np.int64)})
    return out


def from_numpy(data):
    """numpy leaves -> torch tensors, recursively (reference data.py:564-575)."""
    if isinstance(data, dict):
        return {k: from_numpy(v) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [from_numpy(x) for x in data]
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    return data


class Batch(dict):
    """The reference's batch -- a dict of per-scene lists (data.py:555-561) -- that also carries, as an ATTRIBUTE
    (the dict's keys stay exactly the reference's), the same scenes packed for the device: ``flat`` =
    (engine.HostFlatBatch, (actor tracks, rot, orig, actors per scene)).  The packed copy is built on first use
    (Net.forward under no_grad) and kept; ``collate_fn(..., pack=True)`` builds it at collate time, i.e. in the
    DataLoader worker.  Training batches never read it and never pay for it."""
    _scenes = None       # the collated scenes (torch leaves), kept for the lazy pack
    _flat = None
    _flat_tried = False

    @property
    def flat(self):
        if self._flat is None and not self._flat_tried and self._scenes is not None:
            self._flat_tried = True
            self._flat = _pack(self._scenes)
            self._scenes = None
        return self._flat

    @flat.setter
    def flat(self, value):
        self._flat, self._flat_tried = value, True


_pack_warned = [False]


def _pack(scenes):
    """(HostFlatBatch, host actor inputs) of a list of scenes, or None (with ONE warning per process) when the scenes do
    not have the packed layout's fields or the pack fails -- Net.forward then collates the dict-of-lists batch itself."""
    if len(scenes) == 0 or not all(k in scenes[0] for k in ("graph", "feats", "ctrs", "rot", "orig")):
        return None
    try:
        from .engine import collate_flat_host, host_actor_inputs
        # pinned staging only in a process that already talks to the GPU (never initialise it in a loader worker)
        pin = torch.cuda.is_available() and torch.cuda.is_initialized() and torch.utils.data.get_worker_info() is None
        return (collate_flat_host(scenes, pin=pin), host_actor_inputs(scenes))
    except Exception as e:      # noqa: BLE001 -- the packed copy is an accelerator, never a requirement
        if not _pack_warned[0]:
            _pack_warned[0] = True
            import warnings
            warnings.warn("lanegcn_amd.data: packing a batch for the device failed (%r); Net.forward will collate it "
                          "per call instead" % (e,))
        return None


def collate_fn(batch, pack: bool = False):
    """list of scene dicts -> dict of per-key lists, no padding (reference data.py:555-561).  pack=True also builds the
    packed copy (Batch.flat: one staging buffer) here -- in the DataLoader worker when there is one -- so that
    Net.forward(data) under no_grad starts with one host-to-device copy; by default it is built on first use."""
    batch = from_numpy(batch)
    out = Batch({k: [scene[k] for scene in batch] for k in batch[0].keys()})
    out._scenes = batch
    if pack:
        out.flat        # noqa: B018 -- builds and caches it
    return out


def collate_fn_packed(batch):
    """collate_fn for evaluation loaders: the packed copy is built in the loader."""
    return collate_fn(batch, pack=True)


# ---------------------------------------------------------------- synthetic scenes
def _road(rng, L, extent=60.0, spacing=2.0, lat=3.5, n_lat=2):
    """One road: n_lat parallel chains of 9*L segment nodes (centre = midpoint, feat = vector)."""
    n = 9 * L
    th = rng.uniform(0.0, 2.0 * np.pi)
    d = np.array([np.cos(th), np.sin(th)])
    nrm = np.array([-d[1], d[0]])
    org = rng.uniform(-extent, extent, 2)
    pts = org[None, :] + spacing * np.arange(n + 1)[:, None] * d[None, :]
    ctrs, feats = [], []
    for k in range(n_lat):
        p = pts + k * lat * nrm[None, :]
        ctrs.append((p[:-1] + p[1:]) / 2.0)
        feats.append(p[1:] - p[:-1])
    return n, np.concatenate(ctrs, 0), np.concatenate(feats, 0)


def synth_scene(rng, roads, n_actors=50, meta_p=0.3, idx_dtype=np.int64, num_scales=NUM_SCALES):
    """One scene dict in the reference schema.  roads = list of lane counts L (chains of 9*L nodes,
    duplicated into two parallel copies 3.5 m apart linked by left/right edges)."""
    ctrs, feats, pre_u, pre_v, left_u, left_v, lane_idcs = [], [], [], [], [], [], []
    base = 0
    for L in roads:
        n, c, f = _road(rng, L)
        ctrs.append(c)
        feats.append(f)
        for k in range(2):
            i = base + k * n + np.arange(n - 1)
            pre_u.append(i + 1)       # out[u] += W x[v]: predecessor i feeds i + 1
            pre_v.append(i)
        left_u.append(base + np.arange(n))
        left_v.append(base + n + np.arange(n))
        lane_idcs.append(np.repeat(np.arange(2 * L), 9) + (lane_idcs[-1][-1] + 1 if lane_idcs else 0))
        base += 2 * n
    N = base
    pre0 = {"u": np.concatenate(pre_u).astype(np.int64), "v": np.concatenate(pre_v).astype(np.int64)}
    suc0 = {"u": pre0["v"].copy(), "v": pre0["u"].copy()}
    pre = [pre0] + dilated_nbrs(pre0, N, num_scales)
    suc = [suc0] + dilated_nbrs(suc0, N, num_scales)
    lu, lv = np.concatenate(left_u).astype(np.int64), np.concatenate(left_v).astype(np.int64)
    cast = lambda d: {k: v.astype(idx_dtype) for k, v in d.items()}
    graph = dict(
        num_nodes=N,
        ctrs=np.concatenate(ctrs, 0).astype(np.float32),
        feats=np.concatenate(feats, 0).astype(np.float32),
        turn=(rng.random((N, 2)) < meta_p).astype(np.float32),
        control=(rng.random(N) < meta_p).astype(np.float32),
        intersect=(rng.random(N) < meta_p).astype(np.float32),
        pre=[cast(d) for d in pre],
        suc=[cast(d) for d in suc],
        left=cast({"u": lu, "v": lv}),
        right=cast({"u": lv, "v": lu}),
        lane_idcs=np.concatenate(lane_idcs).astype(np.int64),
    )
    a = int(n_actors)
    pick = rng.integers(0, N, a)
    actor_ctrs = (graph["ctrs"][pick] + rng.normal(0.0, 1.0, (a, 2))).astype(np.float32)
    steps = rng.normal(0.0, 0.5, (a, 20, 2))
    actor_feats = np.concatenate([steps, np.ones((a, 20, 1))], 2).astype(np.float32)
    gt = actor_ctrs[:, None, :] + np.cumsum(rng.normal(0.0, 0.5, (a, 30, 2)), 1)
    return dict(
        feats=actor_feats,
        ctrs=actor_ctrs,
        orig=np.zeros(2, np.float32),
        theta=0.0,
        rot=np.eye(2, dtype=np.float32),
        gt_preds=gt.astype(np.float32),
        has_preds=np.ones((a, 30), bool),
        graph=graph,
    )


def synth_batch(kind="S2", seed=0, n_scenes=None, idx_dtype=np.int64):
    """Canonical workloads of SURVEY.md 8(d): S0 (1 scene, 648 nodes), S1 (one merged graph,
    10,008 nodes), S2 (32 scenes x 324 nodes, 50 actors each).  Returns a list of scene dicts."""
    rng = np.random.default_rng(seed)
    if kind == "S0":
        return [synth_scene(rng, [6] * 6, 50, idx_dtype=idx_dtype) for _ in range(n_scenes or 1)]
    if kind == "S1":
        return [synth_scene(rng, [6] * 6 + [1] * 520, 50, idx_dtype=idx_dtype) for _ in range(n_scenes or 1)]
    if kind == "S2":
        return [synth_scene(rng, [6] * 3, 50, idx_dtype=idx_dtype) for _ in range(n_scenes or 32)]
    raise ValueError(kind)


class SyntheticArgoDataset(torch.utils.data.Dataset):
    """Stands in for ArgoDataset(split, config, train) (reference data.py:16-361, needs argoverse-api and
    the dataset, both absent): same constructor signature and item schema, scenes from synth_scene."""

    def __init__(self, split=None, config=None, train=True, length=64, roads=(6, 6, 6), n_actors=50, seed=0):
        if split:      # a real split path was given: say loudly that it is NOT being read
            import warnings
            warnings.warn("SyntheticArgoDataset ignores the split path %r: scenes are random synthetic lane graphs "
                          "(put the real ArgoDataset class in config['dataset_cls'] to train / evaluate on data)" % (split,),
                          RuntimeWarning, stacklevel=2)
        self.config, self.train = config, train
        self.length, self.roads, self.n_actors, self.seed = int(length), list(roads), n_actors, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        scene = synth_scene(np.random.default_rng(self.seed * 1000003 + idx), self.roads, self.n_actors)
        scene["idx"] = idx
        return scene
