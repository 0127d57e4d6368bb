#!/bin/bash
# Developer tool: builds ablated variants of the fused quad kernel (KQ_ABL bits, kns_gruq.hip) into lib/libpv_koala_abl<N>.so
# (developer build) -- timing experiments only, their results are garbage.   tools/quad_abl.sh 1 2 4 ...
cd "$(dirname "$0")/../koala_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-result -DKNS_DEV -DKQ_ABL=$n ${KQ_EXTRA} -x hip -c csrc/kns_gruq.hip -o obj/dev/kns_gruq_abl$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC obj/dev/kns_stft.o obj/dev/kns_gemm.o obj/dev/kns_gru.o obj/dev/kns_gruq_abl$n.o obj/dev/kns_engine.o obj/dev/pv_api.o -shared -Wl,-soname,libpv_koala_dev.so -Wl,--version-script=csrc/libpv_koala.map -o lib/libpv_koala_abl$n.so || exit 1
done
