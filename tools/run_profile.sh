#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summaries -> gpurun_out/profile/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile
mkdir -p $OUT
python $R/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $R/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python $R/tools/gaps.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) 60 > $OUT/timed_region.txt 2>&1
tail -c 2500 $OUT/bench_plain.json
head -12 $OUT/kernel_stats.csv
head -3 $OUT/timed_region.txt
