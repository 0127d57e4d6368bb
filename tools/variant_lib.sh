#!/bin/bash
# Developer tool: a developer-build library with ONE kernel source compiled under extra hipcc flags.
#   tools/variant_lib.sh NAME "-DFLAG=1 ..." [kns_gemm|kns_gru|kns_gruq|kns_stft]   ->  build/ab/libNAME.so
# (the other objects are the tree's developer objects: run `make -C koala_amd` first)
cd "$(dirname "$0")/../koala_amd" || exit 1
name=$1; flags=$2; src=${3:-kns_gemm}
mkdir -p ../build/ab
extra=""; [ "$src" = kns_stft ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off $extra -DKNS_DEV $flags -x hip \
    -c csrc/$src.hip -o ../build/ab/$src.$name.o || exit 1
objs=""
for o in kns_stft kns_gemm kns_gru kns_gruq kns_engine pv_api; do
  if [ $o = $src ]; then objs="$objs ../build/ab/$src.$name.o"; else objs="$objs obj/dev/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC $objs -shared -Wl,--version-script=csrc/libpv_koala.map -o ../build/ab/lib$name.so
