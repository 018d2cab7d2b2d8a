#!/bin/bash
# coverage-map rows for one selection with several libraries (measurement tooling): tools/map_ab.sh "<--only pattern>" [extra coverage_map args]
SEL=$1; shift
for lib in fiat_amd/csrc/libfiat_amd.so build/ab/*.so; do
  echo "== $(basename $lib)"
  FIAT_AMD_LIB=$PWD/$lib python tools/coverage_map.py --only "$SEL" "$@" 2>/dev/null | grep "% HBM" | sort -u | cut -c1-150
done
