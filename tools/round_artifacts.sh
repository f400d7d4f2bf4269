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
python bench.py > "$out/bench_s2.json" 2> "$out/bench_s2.err"
echo "bench rc=$?"; tail -c 600 "$out/bench_s2.json" | head -c 300; echo
python tools/stamps_lc.py 1 f16x2 32 2 2>&1 | grep -v amdgpu.ids > "$out/stamps_lc_tile_short.txt"
for w in a2a m2a a2m; do python tools/stamps_att.py $w 2>&1 | grep -v amdgpu.ids; done > "$out/stamps_att_pairs.txt"
bash tools/pmc_lc.sh short 1:2 > "$out/pmc.log" 2>&1
cp gpurun_out/pmc_short_counters.csv "$out/" 2>/dev/null
bash tools/prof_bench.sh prof_1s --streams 1 --steps 60 --warmup 5 --other-modes "" --no-extras > /dev/null 2>&1
python tools/trace_summary.py gpurun_out/prof_1s/prof_1s_results.db 0.5 "$out/kernel_stats_bench_1stream.csv" > "$out/trace_1stream.txt"
bash tools/prof_bench.sh prof_4s --steps 160 --warmup 10 --other-modes "" --no-extras > /dev/null 2>&1
python tools/trace_summary.py gpurun_out/prof_4s/prof_4s_results.db 0.5 "$out/kernel_stats_bench_4streams.csv" > "$out/trace_4streams.txt"
python tools/trace_concurrency.py gpurun_out/prof_4s/prof_4s_results.db > "$out/trace_4streams_concurrency.txt"
ls -la "$out"
