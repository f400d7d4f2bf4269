#!/bin/bash
# Everything the round's profiles/ files are made from, in one GPU-box call (writes under gpurun_out/final/).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out/final"
mkdir -p "$out"
cd "$root"
# the stamp timelines come from the diagnostic build: it must be at least as new as the shipped library
if [ ! -f lanegcn-1_amd/liblgcn_stamps.so ] || [ lanegcn-1_amd/liblgcn_stamps.so -ot lanegcn-1_amd/liblgcn.so ]; then
    echo "liblgcn_stamps.so is missing or older than liblgcn.so: run 'make -C lanegcn-1_amd/csrc stamps' first" >&2
    exit 1
fi
# 1. the driver's own command, then the steady-state figure (longer run), then the 256-scene batch on one stream
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_s2.json" 2> "$out/bench_s2.err"
echo "bench rc=$?"; tail -c 600 "$out/bench_s2.json" | head -c 300; echo
python bench.py --steps 400 --warmup 40 --other-modes "" --no-extras --cpu-seconds 0 > "$out/bench_s2_steady.json" 2> /dev/null
python bench.py --scenes 256 --streams 1 --steps 30 --warmup 5 --other-modes "" --no-extras --cpu-seconds 0 > "$out/bench_s2x8.json" 2> /dev/null
echo "benches done"
# 2. in-kernel stamp timeline of the LaneConv tile kernel (diagnostic build)
python tools/stamps_lc.py 1 f16x2 32 2 2>&1 | grep -v amdgpu.ids > "$out/stamps_lc_tile_short.txt"
# 3. PMC passes (one counter group per run, never with a trace) over the LaneConv layer
bash tools/pmc_lc.sh short 1:2 > "$out/pmc.log" 2>&1
cp gpurun_out/pmc_short_counters.csv "$out/" 2>/dev/null
# 4. kernel traces of the bench: one forward at a time, four in flight
bash tools/prof_bench.sh prof_1s --streams 1 --steps 60 --warmup 5 --other-modes "" --no-extras --cpu-seconds 0 > /dev/null 2>&1
db=$(find gpurun_out/prof_1s -name "*_results.db" | sort | tail -1)
cp gpurun_out/prof_1s_kernel_stats.csv "$out/kernel_stats_bench_1stream.csv"
python tools/trace_summary.py "$db" 0.5 /dev/null > "$out/trace_1stream.txt"
bash tools/prof_bench.sh prof_4s --steps 160 --warmup 10 --other-modes "" --no-extras --cpu-seconds 0 > /dev/null 2>&1
db=$(find gpurun_out/prof_4s -name "*_results.db" | sort | tail -1)
cp gpurun_out/prof_4s_kernel_stats.csv "$out/kernel_stats_bench_4streams.csv"
python tools/trace_summary.py "$db" 0.5 /dev/null > "$out/trace_4streams.txt"
python tools/trace_concurrency.py "$db" > "$out/trace_4streams_concurrency.txt"
ls -la "$out"
