#!/bin/bash
# rocprofv3 evidence for chosen per-request-cell shapes (run on the GPU box via gpurun; measurement tooling):
#   tools/profile_shapes.sh <tag> "<family,sd,degree,points,order>;..." [instance_ab.py options]  -> gpurun_out/prof_<tag>/..., summary on stdout
# kernel trace + stats, then one PMC pass each for WRITE_SIZE, FETCH_SIZE and the MFMA-busy / GRBM pair (never combined with a trace).
TAG=$1; SHAPES=$2; shift 2
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/instance_ab.py "$SHAPES" "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/instance_ab.py "$SHAPES" "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/instance_ab.py "$SHAPES" "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 tools/instance_ab.py "$SHAPES" "$@" > $OUT/pmc_mfma.log 2>&1
grep "% HBM" $OUT/trace.log
python3 tools/summarize_shapes.py $OUT
