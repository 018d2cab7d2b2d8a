#!/bin/bash
# Sample shader clock / power with rocm-smi while a tabulation loop runs (measurement tooling).
# usage: tools/clock_watch.sh <FIAT_AMD_DEBUG value> [kernel]
DBG=${1:-0}; K=${2:-stream}
FIAT_AMD_KERNEL=$K python tools/kernel_ab.py --variants $DBG --rounds 40 --reps 200 > /tmp/cw_$DBG.log 2>&1 &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | tr '\n' ' ' ; echo
  sleep 0.5
done
wait $PID
tail -1 /tmp/cw_$DBG.log
