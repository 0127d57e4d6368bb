#!/bin/bash
# Developer tool (GPU box): the record kept in profiles/rNN_quad_kernel.txt -- the fused quad GRU-layer kernel against the
# two-kernel form (bit comparison + times), its ablations (tools/quad_abl.sh must have built lib/libpv_koala_abl{1,3,7,15,31}.so)
# and the per-phase stamps of one workgroup.
cd "$(dirname "$0")/.."
echo "# commit $(cat build/head_commit.txt 2>/dev/null)  $(date -u +%FT%TZ)"
echo "## tools/quad_check.py: fused (KOALA_AMD_QUAD=1) vs two-kernel form, developer library"
python tools/quad_check.py 2>&1 | grep -v amdgpu.ids
echo "## ablations of the fused kernel (results garbage; KQ_ABL bits: 1 no remote gather, 2 no x staging loads, 4 no global stores of h, 8 no gate math, 16 no MFMAs)"
(cd koala_amd/lib && python ../../tools/quad_time.py libpv_koala_dev.so libpv_koala_abl1.so libpv_koala_abl3.so libpv_koala_abl7.so libpv_koala_abl15.so libpv_koala_abl31.so 2>&1 | grep -v amdgpu.ids)
echo "## tools/quad_timing.py: s_memtime stamps of workgroup 0 (quad 0, c = 0) and 24 (quad 0, c = 3), shader cycles"
python tools/quad_timing.py 0 64 2>&1 | grep -E "ticks|^x wave|^h wave"
python tools/quad_timing.py 24 64 2>&1 | grep -E "ticks|^x wave|^h wave"
