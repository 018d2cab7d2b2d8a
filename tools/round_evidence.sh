#!/bin/bash
# The round's evidence in one go (run on the GPU box via gpurun): one bench line per BASELINE config with the driver's flags, then the
# rocprofv3 passes of tools/profile.sh for each.  usage: tools/round_evidence.sh <tag> [lines|profiles]
TAG=${1:-r04}; WHAT=${2:-all}
mkdir -p gpurun_out
if [ "$WHAT" != profiles ]; then
  : > gpurun_out/${TAG}_bench_lines.jsonl
  for wl in p3tet c3 n2tet rt2tet dg6tet dg6tet122 hex; do
    python3 bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${TAG}_bench_lines.jsonl
    echo "line $wl done"
  done
fi
if [ "$WHAT" != lines ]; then
  for wl in p3tet n2tet rt2tet dg6tet dg6tet122 hex; do
    bash tools/profile.sh ${TAG}_$wl --workload $wl > gpurun_out/${TAG}_prof_$wl.log 2>&1
    echo "profile $wl done"
  done
fi
