#!/bin/bash
# Developer tool: sustained (power-capped steady state) A/B of library variants through bench.py.
#   tools/ab_bench.sh VARIANT [VARIANT ...]   VARIANT = "-" (the tree as it is) | path/to/kns_gru_variant.hip | path/to/kns_gemm_variant.hip | "-Dflag ..." (extra
#   hipcc flags for the tree's sources)
cd "$(dirname "$0")/.."
mkdir -p build/ab
i=0
for v in "$@"; do
  lib=$PWD/build/ab/libv$i.so
  gru=koala_amd/csrc/kns_gru.hip
  gemm=koala_amd/csrc/kns_gemm.hip
  stft=koala_amd/csrc/kns_stft.hip
  flags=""
  case "$v" in  # "file,flags" = both
    *,*) flags="${v#*,}"; v="${v%%,*}" ;;
  esac
  case "$v" in
    -) ;;
    -*) flags="$v" ;;
    *gemm*) gemm=$v ;;
    *stft*) stft=$v ;;
    *) gru=$v ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Xarch_host -mfma -Xarch_host -mavx2 $flags -Ikoala_amd/csrc -x hip \
      $stft $gemm $gru koala_amd/csrc/kns_gruq.hip koala_amd/csrc/kns_engine.cpp koala_amd/csrc/pv_api.cpp \
      -shared -o $lib || exit 1
  i=$((i+1))
done
for rep in 1 2; do
  for j in $(seq 0 $((i-1))); do
    python bench.py --library $PWD/build/ab/libv$j.so --no-cpu-baseline --steps ${AB_STEPS:-600} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant $j rep $rep: %.2f Mframes/s  %.4f ms/step | ' % (d['value']/1e6, d['ms_per_step']) + '  '.join('%s %.1f' % (k, v['avg_launch_ms']*1e3) for k,v in d['stages'].items()))"
  done
done
