#!/bin/bash
# run one measurement tool with the product library and every library under build/ab/ (measurement tooling): tools/tool_ab.sh <python tool and args>
for lib in fiat_amd/csrc/libfiat_amd.so build/ab/*.so; do
  echo "== $(basename $lib)"
  FIAT_AMD_LIB=$PWD/$lib timeout -k 10 600 python "$@" 2>/dev/null | grep -E "% HBM|%  |us " | cut -c1-170
done
