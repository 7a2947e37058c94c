#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --no-validation > $R/gpurun_out/gaps_bench.log 2>&1
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python $R/tools/gaps.py $f 60 > $R/gpurun_out/gaps.txt 2>&1
tail -3 $R/gpurun_out/gaps_bench.log
cat $R/gpurun_out/gaps.txt
