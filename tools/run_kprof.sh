#!/bin/bash
# rocprofv3 kernel trace of tools/kbench.py (KB_ONLY etc. from the environment); per-kernel stats -> gpurun_out/$1/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-kprof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
rm -rf /tmp/kprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o kb -- python $R/tools/kbench.py > $OUT/kbench.log 2> $OUT/rocprof.err
cp $(find /tmp/kprof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
tail -15 $OUT/kbench.log
grep -E "k_scatter|k_encode4d|Name" $OUT/kernel_stats.csv | cut -c1-200
