#!/bin/bash
# HBM traffic of the macro kernels (separate --pmc passes, as the HBM section of MI355X_MICROARCH.md prescribes):
# tools/pmc_macro.sh <tag> [bench_macro.py arguments]   -> gpurun_out/pmcm_<tag>/traffic.json
TAG=${1:-r1}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/pmcm_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 tools/bench_macro.py --reps 5 "$@" > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % c):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tabulate_macro" in r["Kernel_Name"] and r["Counter_Name"] == c and int(r["Grid_Size"]) > 50000:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = sorted(v)
            res[k][c] = v[len(v) // 2]          # median over the launches (KiB)
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024     # gfx950: FETCH_SIZE counts 64 B per 128-B request
        out[k] = {"read": rd, "write": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
