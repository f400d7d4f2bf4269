#!/bin/bash
# rocprofv3 kernel trace of one bench configuration; summary CSVs land under gpurun_out/<name>/.
# usage (on the GPU box): bash tools/prof_bench.sh <name> [bench.py args...]
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$name" -o "$name" -- python3 "$root/bench.py" "$@" > "$root/gpurun_out/$name.log" 2>&1
# ROCm 7.2 writes a rocpd SQLite file (<name>_results.db), no CSV: tools/rocpd_stats.py makes the per-kernel table
db=$(find "$root/gpurun_out/$name" -name "*_results.db" | sort | tail -1)
python3 "$root/tools/rocpd_stats.py" "$db" "$root/gpurun_out/${name}_kernel_stats.csv" | cut -c1-160 | sed -n 1,30p
