#!/bin/bash
# Developer tool (through gpurun): board power and sclk while ONE variant of build/gru_bench loops.  usage: power_probe.sh <variant indices...>
cd "$(dirname "$0")/../.."
for v in "$@"; do
  build/gru_bench 64 256 10 $v 5 &
  BP=$!
  sleep 2.5
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '
    echo " <- variant $v"
    sleep 0.5
  done
  wait $BP
done
