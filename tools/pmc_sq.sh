#!/bin/bash
# SQ counters of the tabulation kernel for one workload: tools/pmc_sq.sh <tag> <workload> <batch>
TAG=$1; WL=$2; B=$3
export TMPDIR=/tmp
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
CMD="python3 tools/kernel_ab.py --workload $WL --batch $B --variants 0 --rounds 2 --reps 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- $CMD > $OUT/$C.log 2>&1; done
python3 - <<PY
import csv, glob, collections
for d in ("a","b","c","FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tabulate_simplex" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64*256:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print("%-34s n=%d mean=%.4g  per-request=%.1f" % (k, len(v), sum(v)/len(v), sum(v)/len(v)/$B))
PY
tail -2 $OUT/a.log
