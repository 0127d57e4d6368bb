#!/bin/bash
# Developer tool (run through gpurun): board power and clocks (rocm-smi) while the bench workload -- or the kernels of
# one class only, KOALA_AMD_ONLY_CLASS -- loops.   tools/clock_probe.sh [classes...]   (default: all 0 1 2 3 4)
cd "$(dirname "$0")/.."
classes=${@:-all 0 1 2 3 4}
names=(analysis gemm_input gru_recurrent gemm_head synthesis)
for c in $classes; do
  if [ "$c" = all ]; then unset KOALA_AMD_ONLY_CLASS; label=all; steps=6000; else export KOALA_AMD_ONLY_CLASS=$c; label=${names[$c]}; steps=20000; fi
  python bench.py --steps $steps --warmup 3 --no-cpu-baseline > /tmp/bench_probe.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in 1 2 3; do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '
    echo " <- $label"
    sleep 0.5
  done
  kill $BP 2>/dev/null; wait $BP 2>/dev/null
done
