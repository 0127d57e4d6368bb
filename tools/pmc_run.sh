#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one counter group per pass (kernel-trace only, as the
# MI355X guide prescribes), and writes per-kernel sums to gpurun_out/pmc_<tag>/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --prime-seconds 0 --no-cpu-baseline --no-extra --sustain-seconds 0"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.json 2> $OUT/pass$i.err
  python $ROOT/tools/rocpd_summary.py $(ls $OUT/pass$i/*.db | head -1) $OUT/pass$i.txt > /dev/null 2>&1
done
cat $OUT/pass*.txt > $OUT/summary.txt
rm -rf $OUT/pass*/
echo done
