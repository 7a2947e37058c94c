#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries -> gpurun_out/$1/ (default: profile)
# usage (on the GPU box, from the repo root): bash tools/run_profile.sh r02_v1 [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-profile}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py "$@" > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $R/bench.py --no-cpu-baseline --no-validation --curve '' "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python $R/tools/gaps.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) 60 > $OUT/timed_region.txt 2>&1
tail -c 1500 $OUT/bench_plain.json
head -14 $OUT/kernel_stats.csv
head -60 $OUT/timed_region.txt
