#!/bin/bash
# PMC passes (one counter group per rocprofv3 run, never with a trace) over the LaneConv micro-benchmark.
# usage (GPU box): bash tools/pmc_lc.sh <tag> <groups spec of tools/bench_lc.py, e.g. 1:2>
set -e
tag=$1; spec=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out/pmc_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    d="$out/$(echo $grp | cut -d' ' -f1)"
    rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 "$root/tools/bench_lc.py" --profile --groups "$spec" > "$out/run.log" 2>&1
    echo "pass [$grp] done"
done
cd "$root" && python3 tools/pmc_summary.py "gpurun_out/pmc_$tag" "$tag" "$out"/*/ | tail -30
