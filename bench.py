#!/usr/bin/env python
"""Headline benchmark: Argoverse-shaped scenes/s, forward, batch = 32 (~10 k lane nodes) per GPU.

One "step" = one pass of the hot path (graph_gather -> CSR plan -> MapNet -> A2M -> M2M -> M2A ->
A2A) over one 32-scene synthetic batch (workload S2: 10,368 lane nodes, 110,592
edges, 1,600 actors) whose flat input buffers are already resident in HBM.  The forward is captured
once in a hipGraph and replayed per step.  N > 1: one process per GPU (torchrun), every rank runs its
own batch (scenes are independent graphs: no data-path collective), value = all scenes / max time.

Prints ONE JSON line on rank 0 (contract in the task brief) carrying `roofline` for the dominant
kernel (the fused LaneConv layer, k_agg_mlp) and `cpu_baseline` (the oracle = CPU restatement of the
reference on the same batch, timed on this box's host cores, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C = 128
# Dense matrix-core peaks (MI355X_MICROARCH.md).  The split mode spends 6 bf16 products per fp32-grade
# multiply-add, so its effective peak in fp32-equivalent flops is the bf16 peak / 6.
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0 / 6.0, "f16x2": 2500.0 / 3.0, "bf16": 2500.0}
PEAK_NOTE = {"f32": "fp32-input MFMA (v_mfma_f32_32x32x2_f32), 157.3 TF dense",
             "bf16x3": "bf16 MFMA dense 2.5 PF / 6 products of the 3-way split (fp32-grade result)",
             "f16x2": "fp16 MFMA dense 2.5 PF / 3 products of the 2-way split (fp32-grade result)",
             "bf16": "bf16 MFMA dense 2.5 PF"}
PEAK_HBM_GBPS = 8000.0             # HBM3E spec


def measured_traffic(workload, mma, impl="fused"):
    """HBM bytes per LaneConv layer from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in their
    own runs, gfx950 corrections of MI355X_MICROARCH.md applied) committed under profiles/; None when
    no summary for this workload / implementation exists."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get("%s/%s" % (workload, mma), {})
        return rec.get("laneconv_hbm_bytes_per_launch" if impl == "fused" else "laneconv_tiled_hbm_bytes_per_layer")
    except (OSError, ValueError):
        return None


def laneconv_algorithmic(n_nodes, sum_e):
    """flops and bytes of ONE LaneConv layer launch (fp32, s = 4)."""
    flops = 2 * C * C * (n_nodes + sum_e) + 2 * C * C * n_nodes
    byts = 4 * C * n_nodes + 4 * C * sum_e + 8 * sum_e + 16 * 4 * C * C + 4 * C * n_nodes + 16 * C
    return flops, byts


def build_modules(seed, device):
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    torch.manual_seed(seed)
    mods = dict(map_net=M.MapNet(M.config), a2m=M.A2M(M.config), m2m=M.M2M(M.config),
                m2a=M.M2A(M.config), a2a=M.A2A(M.config))
    return {k: m.to(device).eval() for k, m in mods.items()}


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota and by the
    16-core share a one-GPU box grants."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("LGCN_BENCH_MAX_CORES", "16"))))


def cpu_baseline(scenes, actors_cpu, mods, budget_s):
    """The oracle (kind "port": CPU restatement pinned to the reference's own outputs) on the same
    batch, all host cores of this process, bounded to ~budget_s seconds."""
    from oracle import lanegcn_oracle as O            # checker / baseline only
    from lanegcn_amd import data as gen
    sd = {}
    for name, m in mods.items():
        for k, v in m.state_dict().items():
            sd["%s.%s" % (name, k)] = v.detach().cpu()
    tscenes = [gen.from_numpy(s) for s in scenes]
    actor_ctrs = [s["ctrs"] for s in tscenes]

    def one():
        g = O.graph_gather([s["graph"] for s in tscenes])
        return O.hot_path(g, actors_cpu, actor_ctrs, sd)

    cores = host_cores()
    res = {}
    for label, nt, share in (("all", cores, 0.7), ("1t", 1, 0.3)):
        torch.set_num_threads(nt)
        log("cpu baseline: %d thread(s)" % nt)
        with torch.no_grad():
            t0 = time.perf_counter()
            one()                                                   # warm-up
            log("cpu baseline warm-up run %.2f s" % (time.perf_counter() - t0))
            times, t_start = [], time.perf_counter()
            while len(times) < 2 or (time.perf_counter() - t_start < budget_s * share and len(times) < 50):
                t0 = time.perf_counter()
                one()
                times.append(time.perf_counter() - t0)
        res[label] = (float(np.median(times)), len(times))
    n = len(scenes)
    return {
        "value": n / res["all"][0], "unit": "scenes/s", "cores": cores, "kind": "port",
        "sample": "oracle hot path (graph_gather+MapNet+A2M+M2M+M2A+A2A) on the same %d-scene batch, "
                  "median of %d runs at %d torch threads" % (n, res["all"][1], cores),
        "ms_per_batch": res["all"][0] * 1e3,
        "value_1_thread": n / res["1t"][0], "ms_per_batch_1_thread": res["1t"][0] * 1e3,
    }


def stage_table(eng, fb, actors, reps=20):
    """Per-stage time (each stage of the forward captured in its own hipGraph, replayed `reps` times between one
    HIP event pair) next to the stage's ALGORITHMIC flops / bytes of SURVEY.md section 8(d) (fp32, s = 4): what the
    reference's arithmetic would have to move, not what this build moves (it hoists 2/3 of the Att GEMMs)."""
    import torch
    C_, s_ = 128, 4
    N, A, E = fb.n_nodes, fb.n_actors, sum(fb.n_edges)
    st, fns = eng.stage_functions(fb, actors)
    for _, fn in fns:
        fn()
    torch.cuda.synchronize()
    P = [int(p.n_pairs.item()) for p in st["pairs"]]
    lc_f = 2 * C_ * C_ * (N + E) + 2 * C_ * C_ * N
    lc_b = s_ * C_ * N + s_ * C_ * E + 8 * E + 16 * s_ * C_ * C_ + s_ * C_ * N + 16 * C_
    att_f = lambda T, p_: 2 * C_ * C_ * (6 * p_ + 2 * T) + 4 * p_ * C_
    att_b = lambda T, p_: 2 * s_ * C_ * T + 2 * s_ * C_ * p_ + 24 * p_ + s_ * (8 * C_ * C_ + 10 * C_)
    alg = {"index": (0.0, 2 * 8 * E * 2 + 8 * (2 * N + 4 * A) + 8 * sum(P)),
           "map_net": (2 * N * (4 * C_ + 2 * C_ * C_) + 4 * lc_f, 16 * N + s_ * C_ * N + 0.13e6 + 4 * lc_b),
           "a2m": (2 * N * 132 * C_ + 2 * att_f(N, P[0]), 2 * s_ * C_ * N + 16 * N + 132 * s_ * C_ + 2 * att_b(N, P[0])),
           "m2m": (4 * lc_f, 4 * lc_b),
           "m2a": (2 * att_f(A, P[1]), 2 * att_b(A, P[1])),
           "a2a": (2 * att_f(A, P[2]), 2 * att_b(A, P[2]))}
    out = {"pairs": {"a2m": P[0], "m2a": P[1], "a2a": P[2]}}
    side = torch.cuda.Stream()
    alive = []      # every stage graph stays alive to the end: its private pool holds the tensors the later stages read
    for name, fn in fns:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                fn()
        alive.append(g)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        f, b = alg[name]
        out[name] = {"us": us, "alg_GFLOP": f / 1e9, "alg_MB": b / 1e6, "TFLOPs": f / us / 1e6, "GBps": b / us / 1e3,
                     "hbm_frac": b / us / 1e3 / PEAK_HBM_GBPS}
    return out


def laneconv_launch_us(eng, fb, feat_map, feat_m2m, impl, reps=20):
    """Average duration of one LaneConv LAYER without per-launch event overhead: the step's 8 layers (MapNet's 4 +
    M2M's 4, their own weights, the batch's CSR plan) captured back-to-back in one hipGraph and replayed `reps` times
    between ONE HIP event pair on the launch stream.  impl "fused": one k_agg_mlp launch per layer; "tiled": the
    weight-stationary k_lc_tile (+ k_lc_combine when the plan has several unit groups).  The rocprofv3 kernel trace
    (profiles/) reports the same figures; the per-launch event pairs of `kernel_avg_us` add ~5 us each."""
    import torch
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import ops
    g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
    plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices],
                         fb.n_nodes)

    def body():
        M.lane_conv(eng.map_net.fuse, feat_map, plan, fb.num_scales, impl=impl)
        M.lane_conv(eng.m2m.fuse, feat_m2m, plan, fb.num_scales, impl=impl)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        body()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            body()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    launches = 1
    if impl == "tiled":
        lcp = ops.lc_plan(plan)               # the plan lane_conv used (cached on the CSR plan)
        launches = 1 if len(lcp.gstart) == 2 else 2
    return e0.elapsed_time(e1) * 1e3 / (8 * reps), launches


def roofline_of(layer_us, n_nodes, sum_e, mma, workload, impl, launches):
    """`roofline` object of one LaneConv implementation: achieved = ALGORITHMIC bytes / flops of one layer
     / the layer's measured duration.  The binding roof is the one the algorithmic work takes longer on
    in this arithmetic mode (fp32 MFMA-bound on paper; the 16-bit-plane modes, >= 417 TF effective, HBM-bound on
    paper: 69 MB / 8 TB/s = 8.6 us vs 4.3 GFLOP / 833 TF = 5.2 us); both fractions are reported."""
    flops, byts = laneconv_algorithmic(n_nodes, sum_e)
    lc_s = layer_us * 1e-6
    ach, ach_gbs = flops / lc_s / 1e12, byts / lc_s / 1e9
    t_hbm, t_mfma = byts / (PEAK_HBM_GBPS * 1e9), flops / (PEAK_TFLOPS[mma] * 1e12)
    hbm = t_hbm >= t_mfma
    return {
        "bound": "hbm" if hbm else "mfma",
        "achieved": ach_gbs if hbm else ach, "peak": PEAK_HBM_GBPS if hbm else PEAK_TFLOPS[mma],
        "unit": "GB/s" if hbm else "TFLOP/s", "frac": ach_gbs / PEAK_HBM_GBPS if hbm else ach / PEAK_TFLOPS[mma],
        "traffic": measured_traffic(workload, mma, impl),
        "hbm": {"achieved_GBps": ach_gbs, "peak_GBps": PEAK_HBM_GBPS, "frac": ach_gbs / PEAK_HBM_GBPS, "roof_us": t_hbm * 1e6},
        "mfma": {"achieved_TFLOPs": ach, "peak_TFLOPs": PEAK_TFLOPS[mma], "frac": ach / PEAK_TFLOPS[mma],
                 "roof_us": t_mfma * 1e6, "peak_note": PEAK_NOTE[mma], "frac_of_f32_mfma_peak": ach / PEAK_TFLOPS["f32"]},
        "kernel": impl, "launches_per_layer": launches, "avg_launch_us": layer_us,
        "timing": "the step's 8 LaneConv layers captured back-to-back, replayed 20x between one HIP event pair on the "
                  "launch stream; avg_launch_us = per LAYER (all of the layer's launches)",
        "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": byts,
    }


def time_steps(step, steps, warmup, barrier, dev, marks=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
    marks: a list that receives, per timed step, the host time at which the step was issued (seconds from t0) and
    whatever step() returned (run_mode's step returns an event recorded behind the replay on its stream)."""
    from lanegcn_amd import dist as D
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
        if marks is not None:
            marks.append((time.perf_counter() - t0, r))
    torch.cuda.synchronize()
    barrier()
    return D.max_over_ranks(time.perf_counter() - t0, dev)     # slowest rank


def replay_trace(lanes, steps):
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(lanes[0][0]):
        ev0.record()
    evs = []
    t0 = time.perf_counter()
    issued = []
    for k in range(steps):
        st, gj = lanes[k % len(lanes)][:2]
        with torch.cuda.stream(st):
            gj.replay()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        evs.append(e)
        issued.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    done = [ev0.elapsed_time(e) for e in evs]
    S = len(lanes)
    per_lane = [[done[k] - (done[k - S] if k >= S else 0.0) for k in range(j, steps, S)] for j in range(S)]
    return {"steps": steps, "lanes": S, "step_done_ms": [round(x, 4) for x in done],
            "step_issued_host_ms": [round(x, 4) for x in issued],
            "forward_ms_by_lane": [[round(x, 4) for x in l] for l in per_lane],
            "note": "step_done_ms[k]: device time at which step k's forward finished, from an event recorded on the idle "
                    "GPU just before step 0 was issued; forward_ms_by_lane[j][i]: time between consecutive completions "
                    "on lane j (its i-th forward)"}


def run_mode(args, mma, mods, scenes, fb, actors, dev, rank, steps, warmup, trace=False):
    """One arithmetic mode: S forwards in flight (the headline figure) and one forward at a time, both captured in
    hipGraphs.  In the 16-bit-plane modes both run LaneConv on the weight-stationary kernel ("tiled"; at S2 its
    "short" shape: 48-row blocks, one launch per layer); f32 and bf16x3 run the one-launch row-tile kernel
    ("fused").  --laneconv forces either.  DESIGN.md sections 3.3 / 4."""
    from lanegcn_amd import data as gen
    from lanegcn_amd import dist as D
    from lanegcn_amd import ops
    from lanegcn_amd.engine import HotPathEngine, collate_flat
    ops.set_mma(mma)
    S = max(1, args.streams)
    impl_multi = args.laneconv or ("fused" if mma in ("f32", "bf16x3") else "tiled")
    impl_one = args.laneconv or ("fused" if mma in ("f32", "bf16x3") else "tiled")
    eng_multi = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"], lane_impl=impl_multi)
    eng_one = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"], lane_impl=impl_one)
    res = {"laneconv_impl": {"in_flight": impl_multi, "single": impl_one}}
    if args.no_graph:
        step = lambda: eng_one.forward(fb, actors, mapnet_only=args.mapnet_only)
        elapsed = time_steps(step, steps, warmup, D.barrier, dev)
        res.update(ms_per_step=elapsed / steps * 1e3, elapsed=elapsed, streams=1)
        res["laneconv_impl"]["in_flight"] = impl_one
        return res, eng_one
    # the lane streams are created FIRST: torch hands out its pool streams in order and ROCm maps consecutive HIP
    # streams to consecutive hardware queues (4 by default), so these S streams get S distinct queues
    lane_streams = lanes_streams(S)
    lanes, pools = [], []
    for j in range(S):
        sc = scenes if j == 0 else gen.synth_batch(args.workload, seed=100 + rank + 1000 * j, n_scenes=args.scenes)
        fbj = fb if j == 0 else collate_flat(sc, dev)
        aj = actors if j == 0 else torch.randn(fbj.n_actors, C, device=dev).relu()
        torch.cuda.empty_cache()
        r0 = torch.cuda.memory_reserved(dev)
        gj, oj = eng_multi.capture(fbj, aj, mapnet_only=args.mapnet_only)
        torch.cuda.empty_cache()           # what stays reserved is the graph's private pool (every buffer of the forward)
        pools.append((torch.cuda.memory_reserved(dev) - r0) / 2**20)
        lanes.append((lane_streams[j], gj, oj, fbj, aj))
    # one-time costs of a graph belong to its capture, not to the first timed replays: every lane's graph is replayed a
    # few times on the stream it will run on (executable-graph upload, first-touch of its private pool)
    # (round 3) 64 rounds, not 3: after the seconds of host-side setup the device needs ~100 forwards under load before it
    # runs at the rate it then keeps -- measured at the driver's flags (K = 20, W = 5), three runs each: 3 rounds 119-120 k
    # scenes/s, 16: 124-126 k, 32: 129 k, 64: 129-131 k, 128: 128-130 k; steady state (>= 200 steps) 131-132 k in all of
    # them.  The timed region itself is unchanged: W warm-up steps, a synchronisation, exactly K steps, a synchronisation.
    for _ in range(SETTLE_ROUNDS):
        for st, gj in [l[:2] for l in lanes]:
            with torch.cuda.stream(st):
                gj.replay()
    torch.cuda.synchronize()
    counter = [0]

    def step():
        # S independent batches, one captured forward each, replayed round-robin on S streams: step k runs on stream
        # k % S, so up to S forwards overlap on the GPU (every step is still a full batch-32 pass)
        st, gj = lanes[counter[0] % len(lanes)][:2]
        counter[0] += 1
        with torch.cuda.stream(st):
            gj.replay()

    elapsed = time_steps(step, steps, warmup, D.barrier, dev)
    res.update(ms_per_step=elapsed / steps * 1e3, elapsed=elapsed, streams=S)
    res["graph_pool_MB_per_lane"] = [round(x, 1) for x in pools]
    if trace:
        # the same K steps once more, untimed on the host, with a HIP event behind every replay: when each step finished
        # on the device, measured from an event recorded on the idle GPU (attributes the fill / drain of the S lanes)
        res["replay_trace"] = replay_trace(lanes, steps)
        # steady state: the same step over >= 200 steps (the fixed costs of a timed region -- the first kernels of S
        # forwards starting together on an idle chip, the last forwards finishing alone -- weigh 1 / steps)
        n_st = max(200, steps)
        res["steady_ms_per_step"] = time_steps(step, n_st, 0, D.barrier, dev) / n_st * 1e3
    # the range guard's device flag is part of every captured forward: none of them may have tripped
    assert not any(int(l[2]["nonfinite"].item()) for l in lanes if "nonfinite" in l[2]), "non-finite features in mode %s" % mma
    if impl_one == impl_multi:
        g1, o1 = lanes[0][1], lanes[0][2]
    else:
        g1, o1 = eng_one.capture(fb, actors, mapnet_only=args.mapnet_only)
    single = time_steps(g1.replay, steps, warmup, D.barrier, dev)
    assert "nonfinite" not in o1 or int(o1["nonfinite"].item()) == 0
    res["single_ms_per_step"] = single / steps * 1e3
    return res, eng_one


_LANE_STREAMS = []
SETTLE_ROUNDS = 64      # untimed replays of every lane's graph right after its capture (run_mode)


def lanes_streams(n):
    """The streams of the forwards in flight, created once per process: torch hands out streams round-robin over a few
    hardware queues, so streams made later (other modes, the whole-Net figures) can land on the same queue twice; every
    measurement with forwards in flight uses these."""
    import torch
    while len(_LANE_STREAMS) < n:
        _LANE_STREAMS.append(torch.cuda.Stream())
    return _LANE_STREAMS[:n]


def config2_mapnet_s1(mods, dev, reps=30):
    """BASELINE config 2 in the same line: MapNet only (graph_gather + CSR plan + stem + 4 LaneConv) on workload S1, one
    merged lane graph of 10,008 nodes / 59,952 edges, in the headline mode and in exact f32 (the config says fp32): one
    forward per replay, captured in a hipGraph, `reps` replays between one HIP event pair."""
    from lanegcn_amd import data as gen
    from lanegcn_amd import ops
    from lanegcn_amd.engine import HotPathEngine, collate_flat
    prev = ops.get_mma()
    out = {}
    try:
        scenes = gen.synth_batch("S1", seed=100)
        fb = collate_flat(scenes, dev)
        actors = torch.zeros(fb.n_actors, C, device=dev)
        flops, byts = laneconv_algorithmic(fb.n_nodes, sum(fb.n_edges))
        out["workload"] = "S1: MapNet only, %d lane nodes, %d edges, one graph" % (fb.n_nodes, sum(fb.n_edges))
        for mma in (prev, "f32") if prev != "f32" else ("f32",):
            ops.set_mma(mma)
            eng = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"])
            g, _ = eng.capture(fb, actors, mapnet_only=True)
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            # the four LaneConv layers alone
            feat = eng.map_net.stem(fb.node_ctrs, fb.node_feats)
            lc_us, _ = laneconv_launch_us(eng, fb, feat, feat, "fused" if mma in ("f32", "bf16x3") else "tiled")
            out[mma] = {"mapnet_forward_us": us, "mapnet_forwards_per_s": 1e6 / us, "laneconv_layer_us": lc_us,
                        "laneconv_hbm_frac": byts / (lc_us * 1e-6) / (PEAK_HBM_GBPS * 1e9),
                        "laneconv_mfma_frac": flops / (lc_us * 1e-6) / (PEAK_TFLOPS[mma] * 1e12)}
    except Exception as e:      # noqa: BLE001 -- auxiliary figure
        out["error"] = repr(e)[:300]
    finally:
        ops.set_mma(prev)
    return out


def extra_timings(mods, scenes, dev):
    """The reference-facing calls on the same batch (not part of `value`): the drop-in ``Net.forward(data)`` under
    no_grad (host collate of the dict-of-lists batch included) and one training step (forward + loss + backward +
    Adam, reference train.py:179-190)."""
    from lanegcn_amd import data as gen
    from lanegcn_amd import lanegcn as M
    out = {}
    try:
        torch.manual_seed(4321)
        net = M.Net(M.config).to(dev)
        for name in ("map_net", "a2m", "m2m", "m2a", "a2a"):
            getattr(net, name).load_state_dict(mods[name].state_dict())
        batch = gen.collate_fn([gen.from_numpy(s) for s in scenes], pack=True)     # packed in the loader
        net.eval()
        with torch.no_grad():
            for _ in range(3):
                net(batch)
            torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                net(batch)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        out["net_forward_dropin_ms"] = float(np.median(ts)) * 1e3
        # the whole Net (ActorNet + hot path + PredNet + world-frame transform) from device-resident flat inputs, captured
        # in one hipGraph: one forward at a time, and four captured forwards in flight on four streams
        from lanegcn_amd.engine import FullNetEngine, collate_flat
        eng = FullNetEngine(net)
        lanes = []
        for j in range(4):
            sc = scenes                      # same batch, its own device buffers per lane
            fbj = collate_flat(sc)
            feats, rot, orig = eng.actor_inputs(sc)
            g, _ = eng.capture(fbj, feats, rot, orig, [len(x["ctrs"]) for x in sc])
            lanes.append((lanes_streams(4)[j], g, len(sc)))
        for key, nl in (("net_forward_graph_ms", 1), ("net_forward_graph4_ms", 4)):
            def run(n):
                for i in range(n):
                    st, g, _ = lanes[i % nl]
                    with torch.cuda.stream(st):
                        g.replay()
                torch.cuda.synchronize()
            run(8)
            t0 = time.perf_counter()
            run(40)
            out[key] = (time.perf_counter() - t0) / 40 * 1e3
        out["net_forward_graph4_scenes_per_s"] = float(np.mean([l[2] for l in lanes])) / out["net_forward_graph4_ms"] * 1e3
        del lanes, eng
        net.train()
        loss_fn = M.Loss(M.config).to(dev)
        from lanegcn_amd.utils import Optimizer
        opt = Optimizer(net.parameters(), M.config)
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            loss = loss_fn(net(batch), batch)["loss"]
            opt.zero_grad()
            loss.backward()
            opt.step(0.0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out["train_step_ms"] = float(np.median(ts[1:])) * 1e3
    except Exception as e:      # noqa: BLE001 -- auxiliary figures must not take the headline down with them
        out["extra_timings_error"] = repr(e)[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="S2", choices=["S0", "S1", "S2"])
    ap.add_argument("--scenes", type=int, default=None, help="override scenes per batch (S2: 32)")
    ap.add_argument("--mapnet-only", action="store_true",
                    help="step = graph_gather + CSR plan + MapNet only (BASELINE config 'MapNet LaneConv only', use with S1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--streams", type=int, default=4,
                    help="forward graphs kept in flight on separate HIP streams (each on its own batch)")
    ap.add_argument("--mma", default=None, choices=["f32", "bf16x3", "f16x2", "bf16"],
                    help="matrix-core mode (default: LGCN_MMA or f16x2 = fp32-grade 2-way fp16 split)")
    ap.add_argument("--laneconv", default=None, choices=["fused", "tiled"],
                    help="force one LaneConv implementation (default: fused with several forwards in flight, tiled for one)")
    ap.add_argument("--other-modes", default="f32,bf16x3,bf16",
                    help="arithmetic modes reported next to the headline one in `modes` (comma list, '' = none)")
    ap.add_argument("--no-extras", action="store_true", help="skip the drop-in Net.forward / training-step timings")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torchrun (one process per GPU)" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data as gen
    from lanegcn_amd import dist as D
    from lanegcn_amd import ops
    if world > 1:
        import torch.distributed as tdist
        tdist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    from lanegcn_amd.engine import collate_flat

    if args.mma:
        ops.set_mma(args.mma)
    mma = ops.get_mma()
    mods = build_modules(1234, dev)            # same random-init weights on every rank
    scenes = gen.synth_batch(args.workload, seed=100 + rank, n_scenes=args.scenes)
    fb = collate_flat(scenes, dev)
    actors_cpu = torch.from_numpy(
        np.random.default_rng(7 + rank).normal(0, 1, (fb.n_actors, C)).astype(np.float32)).relu()
    actors = actors_cpu.to(dev)
    log("rank %d: batch ready (N=%d nodes, A=%d actors, sumE=%d)" % (rank, fb.n_nodes, fb.n_actors, sum(fb.n_edges)))

    head, eng = run_mode(args, mma, mods, scenes, fb, actors, dev, rank, args.steps, args.warmup, trace=(world == 1))
    elapsed = head["elapsed"]
    log("rank %d: %d steps in %.4f s (%s)" % (rank, args.steps, elapsed, mma))

    # per-kernel durations (HIP events on the launch stream), eager launches of the single-forward engine
    with ops.kernel_timer() as kt:
        for _ in range(10):
            eng.forward(fb, actors, mapnet_only=args.mapnet_only)
    ksum = kt.summary()
    lc_us, lc_launches, stages_tab, modes, extras = {}, {}, None, {}, {}
    n_scenes = len(scenes)
    if rank == 0:
        st = eng.forward(fb, actors, stages=not args.mapnet_only, mapnet_only=args.mapnet_only)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in st.values() if torch.is_tensor(v) and v.is_floating_point())
        f_map = st["nodes"] if args.mapnet_only else st["map_net"]
        f_m2m = st["nodes"] if args.mapnet_only else st["a2m"]
        for impl in (("fused",) if mma in ("f32", "bf16x3") else ("fused", "tiled")):      # tiled: two- / one-plane modes
            lc_us[impl], lc_launches[impl] = laneconv_launch_us(eng, fb, f_map, f_m2m, impl)
        stages_tab = None if args.mapnet_only or world > 1 else stage_table(eng, fb, actors)
    if world == 1 and not args.mapnet_only and not args.no_graph:
        for m in [v for v in args.other_modes.split(",") if v and v != mma]:
            r, e_m = run_mode(args, m, mods, scenes, fb, actors, dev, rank, max(40, args.steps // 2), args.warmup)
            stm = e_m.forward(fb, actors, stages=True)
            lcm = {impl: laneconv_launch_us(e_m, fb, stm["map_net"], stm["a2m"], impl)[0]
                   for impl in (("fused",) if m in ("f32", "bf16x3") else ("fused", "tiled"))}
            dom = r["laneconv_impl"]["in_flight"]
            modes[m] = {
                "value": n_scenes / (r["ms_per_step"] * 1e-3), "unit": "scenes/s", "ms_per_step": r["ms_per_step"],
                "streams": r["streams"], "single_stream_value": n_scenes / (r["single_ms_per_step"] * 1e-3),
                "single_stream_ms_per_step": r["single_ms_per_step"], "laneconv_impl": r["laneconv_impl"],
                "laneconv_layer_us": lcm,
                "roofline": {k: roofline_of(lcm[dom], fb.n_nodes, sum(fb.n_edges), m, args.workload, dom, 1)[k]
                             for k in ("bound", "achieved", "peak", "unit", "frac", "hbm", "mfma")},
            }
            log("mode %s: %.0f scenes/s" % (m, modes[m]["value"]))
        ops.set_mma(mma)
        if not args.no_extras:
            extras = extra_timings(mods, scenes, dev)
            extras["config2_mapnet_s1"] = config2_mapnet_s1(mods, dev)

    if rank == 0:
        sum_e = sum(fb.n_edges)
        how = "eager" if args.no_graph else "hipGraph replay" + (
            "" if head["streams"] <= 1 else ", %d forwards in flight on %d streams" % (head["streams"], head["streams"]))
        what = ("MapNet only (graph_gather+CSR plan+stem+4 LaneConv)" if args.mapnet_only
                else "hot path forward (graph_gather+CSR plan+MapNet+A2M+M2M+M2A+A2A)")
        workload_desc = ("%s: %s, %d scenes/GPU, %d lane nodes, %d edges, %d actors, random-init weights, inputs "
                         "resident in HBM, %s" % (args.workload, what, n_scenes, fb.n_nodes, sum_e, fb.n_actors, how))
        names = {"fused": ("lgcn::k_agg_mlp<1>" if mma == "f32" else "lgcn::k_agg_mlp_bf<RB,F,1>") + " (one-launch LaneConv layer)",
                 "tiled": "lgcn::k_lc_tile<F,V,FIN>" + (" + lgcn::k_lc_combine<F>" if lc_launches.get("tiled", 1) > 1 else "") +
                          " (weight-stationary LaneConv layer)"}
        dom = head["laneconv_impl"]["in_flight"]
        roofline = roofline_of(lc_us[dom], fb.n_nodes, sum_e, mma, args.workload, dom, lc_launches[dom])
        roofline["kernel"] = names[dom] + ", 8 layers/step: the LaneConv of the timed configuration"
        roofline["avg_launch_us_eager_event_pairs"] = float(np.mean(ksum["laneconv"])) * 1e3
        roofline["note"] = ("traffic = HBM bytes per layer from the PMC passes (profiles/pmc_traffic.json); it is below the "
                            "algorithmic bytes because the gathers of a row hit L2; what binds the one-launch kernel is the "
                            "L2 -> CU weight stream, what binds the weight-stationary one is MFMA issue (DESIGN.md section 4)")
        line = {
            "metric": "Argoverse scenes/sec forward (batch=32, ~10k lane nodes)",
            "value": args.gpus * n_scenes * args.steps / elapsed,
            "unit": "scenes/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (3 bf16 planes per operand, 6 products, fp32 accumulate)",
                      "f16x2": "f32 (2 fp16 planes per operand, 3 products, fp32 accumulate; range-guarded)",
                      "bf16": "bf16"}[mma],
            "mma": mma, "data": "synthetic",
            "config": {"workload": workload_desc,
                       "scenes_per_gpu": n_scenes, "parallelism": "dp%d (independent scene shards)" % args.gpus},
            "roofline": roofline,
            "laneconv": {"impl": head["laneconv_impl"], "layer_us": lc_us, "launches_per_layer": lc_launches,
                         "roofline_by_impl": {k: {kk: roofline_of(v, fb.n_nodes, sum_e, mma, args.workload, k, 1)[kk]
                                                  for kk in ("bound", "achieved", "peak", "unit", "frac")}
                                              for k, v in lc_us.items()}},
            "kernel_avg_us": {k: float(np.mean(v)) * 1e3 for k, v in ksum.items()},
            "streams": head["streams"],
        }
        if stages_tab is not None:
            line["stages"] = stages_tab
        if "graph_pool_MB_per_lane" in head:      # device memory a captured forward keeps (its private pool), per lane
            line["graph_pool_MB_per_lane"] = head["graph_pool_MB_per_lane"]
        if "steady_ms_per_step" in head:
            line["value_steady"] = args.gpus * n_scenes / (head["steady_ms_per_step"] * 1e-3)
            line["steady_ms_per_step"] = head["steady_ms_per_step"]
            line["value_note"] = ("value: exactly --steps steps between two synchronisations (the driver's K), after --warmup "
                                  "steps; before those, every lane's captured graph is replayed %d times (the device reaches "
                                  "its sustained rate only after ~100 forwards under load: with 3 such replays the K = 20 "
                                  "figure is 119-120 k, with 64 it is 129-131 k, DESIGN.md section 4).  value_steady: the same "
                                  "step over max(200, K) steps.  A timed region starts on an idle chip and ends when the last "
                                  "forward has drained: those fixed costs weigh 1 / K (replay_trace)" % SETTLE_ROUNDS)
            line["settle_replays_per_lane"] = SETTLE_ROUNDS
            line["replay_trace"] = head["replay_trace"]
        if "single_ms_per_step" in head:
            line["single_stream"] = {"value": args.gpus * n_scenes / (head["single_ms_per_step"] * 1e-3), "unit": "scenes/s",
                                     "ms_per_step": head["single_ms_per_step"], "laneconv_impl": head["laneconv_impl"]["single"]}
        if modes:
            line["modes"] = modes
        line.update(extras)
        if args.cpu_seconds > 0 and world == 1:      # reported baseline: rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(scenes, actors_cpu, mods, args.cpu_seconds)
            line["speedup_vs_cpu_all_cores"] = line["value"] / args.gpus / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
