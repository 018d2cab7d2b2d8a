#!/bin/bash
# ABAB of in-kernel chain-rule variants of the stacked kernel over the triangle instances, orders 1 and 2 (measurement tooling):
# product library against every library under build/ab/ (e.g. -DFX_MIXR1=2 -DFX_MIXR_WPS=2: two waves per SIMD)
S=""
for o in 1 2; do
S="$S;Nedelec,2,3,12,$o;Nedelec,2,3,16,$o;Nedelec,2,3,22,$o;Nedelec,2,4,16,$o;Nedelec,2,4,22,$o;Nedelec,2,4,30,$o;Lagrange,2,5,24,$o;Lagrange,2,5,30,$o;Lagrange,2,5,44,$o;Lagrange,2,6,23,$o;Lagrange,2,6,30,$o;Lagrange,2,6,44,$o;Lagrange,2,6,33,$o;Lagrange,2,4,16,$o"
done
S=${S#;}
for rep in 1 2; do
for lib in fiat_amd/csrc/libfiat_amd.so build/ab/*.so; do
  echo "== $lib"
  FIAT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tools/instance_ab.py "$S" ${POLICY:+--policy $POLICY} 2>&1 | grep -E "% HBM|Error|error"
done; done
