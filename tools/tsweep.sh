#!/bin/bash
# Developer tool (through gpurun): frames per call sweep of the bench workload.
cd "$(dirname "$0")/.."
for T in 8 16 32 64 128; do
  python bench.py --frames $T --no-cpu-baseline --no-extra --sustain-seconds 0 --steps $((12800 / T)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T=$T: %.2f Mframes/s  %.4f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done
python bench.py --precision fp32 --frames 32 --no-cpu-baseline --no-extra --sustain-seconds 0 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp32 B=4096 T=32: %.2f Mframes/s  %.4f ms/step' % (d['value']/1e6, d['ms_per_step']))"
