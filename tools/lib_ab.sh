#!/bin/bash
# Ablation of the specialised kernel through the FX_DBG builds (make -C fiat_amd/csrc dbg-libs):
# every library is timed with and without the epilogue (FIAT_AMD_DEBUG bit 4).
# usage: tools/lib_ab.sh [kernel_ab.py arguments]
for d in "" ${LIBS:-_dbg1 _dbg2 _dbg3 _dbg8}; do
  lib=fiat_amd/csrc/libfiat_amd$d.so
  [ -f $lib ] || continue
  echo "== $lib"
  FIAT_AMD_LIB=$PWD/$lib timeout -k 10 200 python tools/kernel_ab.py --variants 0,4 --rounds 3 "$@" 2>&1 | tail -2
done
