#!/bin/bash
# Developer tool (through gpurun): the record of the wavefront route -- tools/wave_check.py over batch sizes, frames per call and
# m-tiles per workgroup, then the in-kernel stamps (tools/wave_timing.py).   tools/wave_sweep.sh > gpurun_out/wavefront.txt
cd "$(dirname "$0")/.."
run() { timeout 300 python tools/wave_check.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "# Wavefront route for multi-frame calls of few streams (kns_engine.cpp kRouteWave, kns_gru.hip gru_wave_kernel) against the"
echo "# layer-by-layer routes (fp32: gru_small_kernel frame by frame / chunked kernels; bf16: input GEMM + resident recurrent kernel)."
echo "# tools/wave_check.py, one MI355X box, device-resident PCM, 50 calls timed after 5; 'max |diff|' = the two routes' PCM over"
echo "# three consecutive calls (state carried).  Commit $(git rev-parse --short HEAD 2>/dev/null || echo '?')."
echo
echo "## 32 frames per call, wavefront forced at every size (the engine takes it in fp32 at every size up to 4 096 streams, in bf16 up to 768)"
WAVE_T=32 run 16 64 128 256 512 1024 2048 4096
echo
echo "## 256 and 512 streams, frames per call swept"
for T in 2 4 8 16 64; do WAVE_T=$T run 256 512; done
echo
echo "## 256 streams x 32 frames, m-tiles per workgroup (KOALA_AMD_WAVE_GROUP; the engine's choice at 256 streams: 3)"
for g in 1 2 3 4 6 8; do echo "group $g"; KOALA_AMD_WAVE_GROUP=$g run 256; done
echo
echo "## in-kernel stamps (cycles), fp32 then bf16"
timeout 300 python tools/wave_timing.py 256 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|\^"
WAVE_PREC=bf16 timeout 300 python tools/wave_timing.py 256 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|\^"
