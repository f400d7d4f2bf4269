#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: bash tools/prof_cmd.sh <name> <script.py> [args...]
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
script="$root/$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$name" -o "$name" -- python3 "$script" "$@" > "$root/gpurun_out/$name.log" 2>&1
