#!/bin/bash
# Developer tool (through gpurun): MFMA / VALU co-execution counters of the bench workload for one or more library builds.
#   tools/pmc_coexec.sh TAG lib1.so [lib2.so ...]   ->  gpurun_out/pmc_coexec_TAG.txt (per-kernel sums per dispatch, one block per library)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_coexec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  CMD="python $ROOT/bench.py --library $ROOT/$lib --steps 3 --warmup 1 --prime-seconds 0 --no-cpu-baseline --no-extra --sustain-seconds 0"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS \
      -d $OUT/$name -o p -- $CMD > $OUT/$name.json 2> $OUT/$name.err
  echo "== $lib" >> $OUT.txt
  python $ROOT/tools/rocpd_summary.py $(ls $OUT/$name/*.db | head -1) $OUT/$name.txt > /dev/null 2>&1
  grep "gru_resident8" $OUT/$name.txt >> $OUT.txt
  rm -rf $OUT/$name
done
cat $OUT.txt
