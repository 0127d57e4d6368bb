#!/bin/bash
# Developer tool (run through gpurun): samples rocm-smi clocks/power while the bench workload loops.
cd "$(dirname "$0")/.."
python bench.py --steps ${1:-3000} --warmup 3 --no-cpu-baseline > /tmp/bench_probe.json 2>/dev/null &
BP=$!
sleep ${2:-12}
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ';'
  echo
  sleep 0.7
done
wait $BP
cat /tmp/bench_probe.json | head -c 600
