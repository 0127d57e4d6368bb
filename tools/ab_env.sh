#!/bin/bash
# Developer tool (through gpurun): sustained A/B of environment switches through bench.py.  usage: ab_env.sh "VAR=1" "" "VAR2=x" ...
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for e in "$@"; do
    env $e python bench.py --library $PWD/koala_amd/lib/libpv_koala_dev.so --no-cpu-baseline --no-extra --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] rep $rep: %.2f Mframes/s  %.4f ms/step | ' % ('$e', d['value']/1e6, d['ms_per_step']) + '  '.join('%s %.1f' % (k, v['avg_launch_ms']*1e3) for k,v in d['stages'].items()))"
  done
done
