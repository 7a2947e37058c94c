#!/bin/bash
# round 5, call W: rocprofv3 kernel trace + stats of one trial of the bench on the final tree (the timed region = the last 60 steps).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5w
mkdir -p $OUT
cd $R
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --trials 1 --steps 60 --warmup 20 --no-cpu-baseline --no-validation --curve '' > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python $R/tools/gaps.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) 60 > $OUT/timed_region.txt 2>&1
head -64 $OUT/timed_region.txt | cut -c1-150
python - <<PY
import json
d = json.loads(open("$OUT/bench_under_rocprof.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["samples_per_ray_post"], d["kernel_ms_per_step"], d["roofline"]["frac"], [k["frac"] for k in d["roofline_kernels"]], d["prune_march_launches_per_step"])
PY
