#!/bin/bash
# quick HBM-traffic check of the tabulation kernel: tools/pmc_quick.sh <tag> [env assignments...]
TAG=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 tools/kernel_ab.py --variants 0 --rounds 2 --reps 3 > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % c):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tabulate_simplex" in r["Kernel_Name"] and r["Counter_Name"] == c and int(r["Grid_Size"]) > 100000:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print("$TAG", c, k, "n=%d mean=%.4g KiB -> %.1f MB" % (len(v), sum(v)/len(v), sum(v)/len(v)*1024/1e6))
PY
