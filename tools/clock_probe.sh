#!/bin/bash
# Developer tool (run through gpurun): RAW rocm-smi samples (board power, sclk / mclk) while the bench workload -- or the
# kernels of one class only (KOALA_AMD_ONLY_CLASS, developer library) -- loops, plus the bench's own per-class launch
# times of the same run.   tools/clock_probe.sh [classes...]   (default: all 0 1 2 3 4)
# Output on stdout: one block per class; redirect into gpurun_out/ and copy the file into profiles/rNN_power.txt.
cd "$(dirname "$0")/.."
classes=${@:-all 0 1 2 3 4}
names=(analysis gemm_input gru_recurrent gemm_head synthesis)
echo "# $(date -u +%FT%TZ) commit $(cat build/head_commit.txt 2>/dev/null) ; power cap: $(rocm-smi --showmaxpower 2>/dev/null | grep -i 'max' | sed 's/.*: //' | tr '\n' ' ')"
for c in $classes; do
  if [ "$c" = all ]; then unset KOALA_AMD_ONLY_CLASS; label=all; steps=${PROBE_STEPS_ALL:-5000}; else export KOALA_AMD_ONLY_CLASS=$c; label=${names[$c]}; steps=${PROBE_STEPS_ONE:-20000}; case $c in 0|3|4) steps=$((steps*8));; esac; fi
  python bench.py --library $PWD/koala_amd/lib/libpv_koala_dev.so --steps $steps --warmup 3 --no-cpu-baseline --no-extra --sustain-seconds 0 > /tmp/bench_probe.json 2>/dev/null &
  BP=$!
  sleep ${PROBE_SETTLE:-9}
  echo "== $label"
  for i in 1 2 3 4; do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
    echo
    sleep 0.5
  done
  if [ "$c" = all ]; then
    wait $BP 2>/dev/null
    python - <<'PY'
import json
try:
    d = json.loads(open('/tmp/bench_probe.json').read().strip().splitlines()[-1])
    print('bench: %.2f M frames/s, %.4f ms/step | ' % (d['value'] / 1e6, d['ms_per_step']) +
          '  '.join('%s %.1f us x%s' % (k, v['avg_launch_ms'] * 1e3, v['launches_per_step']) for k, v in d['stages'].items()))
except Exception as e:
    print('bench line unavailable:', e)
PY
  else
    kill $BP 2>/dev/null; wait $BP 2>/dev/null
  fi
done
