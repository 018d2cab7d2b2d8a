#!/bin/bash
# A/B of work-queue variants (measurement tooling): every library in build/ab/ and the product library time the same
# workloads in separate processes, twice, interleaved.  usage: tools/queue_ab.sh "<workload:batches> ..."
SPECS=${1:-"rt2tet:25000 n2tet:25000 p3tet:100000"}
for round in 1 2; do
  for lib in fiat_amd/csrc/libfiat_amd.so build/ab/*.so; do
    for spec in $SPECS; do
      wl=${spec%%:*}; b=${spec##*:}
      printf "%s round %d %s " "$(basename $lib)" $round "$wl"
      FIAT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tools/batch_scaling.py --workload $wl --batches $b --rounds 7 2>/dev/null | grep "batch " 
    done
  done
done
