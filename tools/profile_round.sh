#!/bin/bash
# One-shot round profile on the GPU box: rocprofv3 kernel stats of bench.py, PMC passes, and the bench line itself.
# Outputs land in gpurun_out/round_<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --sustain-seconds 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python $ROOT/tools/rocpd_summary.py $(ls $OUT/stats/*.db | head -1) $OUT/kernel_stats.txt > /dev/null
rm -rf $OUT/stats
bash $ROOT/tools/pmc_run.sh $TAG > /dev/null 2>&1
python $ROOT/tools/pmc_to_json.py $ROOT/gpurun_out/pmc_$TAG $OUT/pmc.json "$(cat $ROOT/build/head_commit.txt 2>/dev/null)" > /dev/null
cp $ROOT/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
cd $ROOT && cp $OUT/pmc.json profiles/${TAG}_pmc.json && python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
