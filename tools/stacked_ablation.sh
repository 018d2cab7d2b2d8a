# stacked-matrix kernel with parts switched off (measurement tooling; make -C fiat_amd/csrc dbg-libs DBGS="1 2 3" first):
# dbg1 no recurrence, dbg2 no MFMAs, dbg3 neither
for lib in "" dbg1 dbg2 dbg3; do
  if [ -n "$lib" ]; then export FIAT_AMD_LIB=$PWD/fiat_amd/csrc/libfiat_amd_$lib.so; else unset FIAT_AMD_LIB; fi
  echo "== lib [$lib]"
  python tools/stacked_ablation.py 2>&1 | grep "order\|Error"
done
