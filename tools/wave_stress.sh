#!/bin/bash
# Developer tool (through gpurun): the wavefront route against the layer-by-layer routes over and over -- seven shapes x six
# repetitions x both precisions, three calls each; any differing sample is reported (the one-barrier form with an LDS flag failed
# 3 of 26 such runs; the committed two-barrier form: 0 of 84).
cd "$(dirname "$0")/.."
n=0; bad=0
for rep in 1 2 3 4 5 6; do
  for cfg in "256 32 0" "256 32 4" "100 17 0" "512 8 0" "1024 12 2" "48 64 1" "2048 24 0"; do
    set -- $cfg
    out=$(KOALA_AMD_WAVE_GROUP=$3 WAVE_T=$2 timeout 300 python tools/wave_check.py $1 2>&1 | grep "max |diff|")
    n=$((n+2))
    if echo "$out" | grep -qv "max |diff| 0,"; then bad=$((bad+1)); echo "DIFF: $cfg"; echo "$out"; fi
  done
done
echo "runs (precision x config x repetition): $n, with a difference: $bad"
