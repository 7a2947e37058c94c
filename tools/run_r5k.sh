#!/bin/bash
# round 5, call K: the driver's command; rocprofv3 kernel trace of the bench (timed region).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5k
mkdir -p $OUT
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_driver_cmd.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "min/max", d["value_min"], d["value_max"])
    for t in d["trials"]: print(t)
    print({k: d[k] for k in ("kernel_ms_per_step", "samples_rendered_per_s", "ms_per_640k_samples", "validation_psnr_db", "setup_s")})
    print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_over_algorithmic"])
    print([ (k["kernel"][:20], k["frac"], k["traffic_over_algorithmic"], k["ms_per_step"]) for k in d["roofline_kernels"]])
    print(d["cpu_baseline"]); print(d["regime_curve"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_driver_cmd.err").read()[-2000:])
PY
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --trials 1 --steps 60 --warmup 20 --no-cpu-baseline --no-validation --curve '' > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python $R/tools/gaps.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) 60 > $OUT/timed_region.txt 2>&1
head -62 $OUT/timed_region.txt
