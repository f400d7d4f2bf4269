"""Summarise rocprofv3 --pmc output directories into the CSV / JSON files kept under profiles/.

Each counter group is collected in its OWN run (never together with a trace):
    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d DIR_F -- python bench.py --steps 20 --warmup 2 --cpu-seconds 0 --no-graph
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d DIR_W -- python bench.py ... (same command)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR_S -- ...
    rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d DIR_T -- ...
Usage: python tools/pmc_summary.py OUT_PREFIX TAG DIR [DIR ...]
writes OUT_PREFIX_counters.csv (kernel, counter, launches, mean) and prints the HBM bytes per launch of the
LaneConv kernel, (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of a 16 B/lane stream,
MI355X_MICROARCH.md, HBM section), as JSON for profiles/pmc_traffic.json under key TAG."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_prefix, tag, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: [0, 0.0])          # (kernel, counter) -> [launches, sum]
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)    # a counter may be split over several rows (per XCD / instance)
            for r in csv.DictReader(open(f)):
                per_dispatch[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
            for (k, c, _), v in per_dispatch.items():
                acc[(k, c)][0] += 1
                acc[(k, c)][1] += v
    rows = sorted(acc.items())
    with open(out_prefix + "_counters.csv", "w") as fo:
        fo.write("kernel,counter,launches,mean\n")
        for (k, c), (n, s) in rows:
            fo.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))
    lc = [k for (k, c) in acc if "k_agg_mlp" in k and ", 1, false>" in k or "k_agg_mlp<1>" in k]
    res = {}
    if lc:
        k = sorted(set(lc), key=lambda kk: -acc.get((kk, "FETCH_SIZE"), [0, 0])[0])[0]
        f = acc.get((k, "FETCH_SIZE"))
        w = acc.get((k, "WRITE_SIZE"))
        if f and w:
            fk, wk = f[1] / f[0], w[1] / w[0]
            res = {tag: {"laneconv_hbm_bytes_per_launch": (2 * fk + wk) * 1024, "kernel": k, "FETCH_SIZE_KB": fk,
                         "WRITE_SIZE_KB": wk,
                         "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of 16 B/lane streams)",
                         "source": os.path.basename(out_prefix) + "_counters.csv"}}
        for c in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            v = acc.get((k, c))
            if v and tag in res:
                res[tag][c] = v[1] / v[0]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
