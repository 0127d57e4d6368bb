#!/bin/bash
# Developer tool: build variant libraries with -D switches and time the kernel classes of each (run through gpurun).
#   tools/ab.sh "" -DR8X_NO_HSEQ "-DR8X_NO_T16 -DR8X_NO_GI"
cd "$(dirname "$0")/.."
mkdir -p build/ab
i=0
for flags in "$@"; do
  lib=build/ab/lib$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off $flags -x hip \
      koala_amd/csrc/kns_stft.hip koala_amd/csrc/kns_gemm.hip koala_amd/csrc/kns_gru.hip koala_amd/csrc/kns_engine.cpp \
      koala_amd/csrc/pv_api.cpp -shared -o $lib || exit 1
  echo "== variant $i: '$flags'"
  SWEEP_LIB=$PWD/$lib SWEEP_T=${AB_T:-32} python tools/sweep.py 2>&1 | grep "^T="
  i=$((i+1))
done
