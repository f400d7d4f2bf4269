#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/prof_train" -o prof_train -- python3 "$root/tools/bench_train.py" --steps 10 --warmup 3 --cpu-steps 0 > "$root/gpurun_out/prof_train.log" 2>&1
db=$(find "$root/gpurun_out/prof_train" -name "*_results.db" | sort | tail -1)
python3 "$root/tools/rocpd_stats.py" "$db" "$root/gpurun_out/prof_train_kernel_stats.csv" | cut -c1-150 | sed -n 1,45p
python3 "$root/tools/trace_summary.py" "$db" 0.04 > "$root/gpurun_out/prof_train_tail.txt" 2>&1
rm -rf "$root/gpurun_out/prof_train"       # the trace database of a training run is > 64 MiB: only the table travels back
