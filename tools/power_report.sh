#!/bin/bash
# Developer tool (GPU box): the record kept in profiles/rNN_power.txt -- RAW rocm-smi samples (board power, sclk) while
#   (1) the bench workload and each kernel class alone loop (tools/clock_probe.sh, developer library, KOALA_AMD_ONLY_CLASS),
#   (2) the recurrent kernel's microbench (build/gru_bench: us per launch, cycles per step of every variant; then variants 0 and 1
#       looping under tools/microbench/power_probe.sh).
cd "$(dirname "$0")/.."
echo "##### (1) bench workload and kernel classes, product kernels"
bash tools/clock_probe.sh all 0 1 2 3 4
echo "##### (2) build/gru_bench 64 256 20: recurrent kernel variants alone (us per launch, s_memtime cycles per step)"
build/gru_bench 64 256 20 2>&1 | tail -12
echo "##### (2b) tools/microbench/power_probe.sh 0 1"
bash tools/microbench/power_probe.sh 0 1
