for rep in 1 2; do
  for lib in build/libpv_koala_mix.so koala_amd/lib/libpv_koala.so; do
    python bench.py --library $PWD/$lib --no-cpu-baseline --no-extra --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rep $rep: %.2f Mframes/s  %.4f ms/step | ' % (d['value']/1e6, d['ms_per_step']) + '  '.join('%s %.1f' % (k, v['avg_launch_ms']*1e3) for k,v in d['stages'].items()))"
  done
done
