#!/bin/bash
# Developer tool (through gpurun): sustained A/B of library builds through bench.py on ONE box in ONE call.
#   tools/ab.sh libA.so libB.so [...]      (paths relative to the repository root; AB_REPS repetitions, default 2)
# Box-to-box spread of the bench is +-2.5 %, run-to-run on one box +-0.3 %: only compare numbers from the same call.
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${AB_REPS:-2}); do
  for lib in "$@"; do
    python bench.py --library $PWD/$lib --no-cpu-baseline --no-extra --sustain-seconds 0 --steps ${AB_STEPS:-400} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rep $rep: %.2f Mframes/s  %.4f ms/step | ' % (d['value']/1e6, d['ms_per_step']) + '  '.join('%s %.1f' % (k, v['avg_launch_ms']*1e3) for k,v in d['stages'].items()))"
  done
done
