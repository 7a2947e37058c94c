#!/bin/bash
# Bench lines of the other single-GPU configurations. usage: bash tools/run_configs.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-configs}
mkdir -p $OUT
COMMON="--no-cpu-baseline --curve '' --validation-views 1 --pretrain 1000 --steps 40"
eval timeout 240 python $R/bench.py $COMMON --partitioning none > $OUT/partitioning_none.json 2> $OUT/partitioning_none.err
eval timeout 300 python $R/bench.py $COMMON --partitioning fixed --segment-size 100 --frames 100 --capture-budget-gb 0 --replacements-per-step 0 > $OUT/segment100_frames100.json 2> $OUT/segment100_frames100.err
eval timeout 400 python $R/bench.py $COMMON --frames 250 --capture-budget-gb 0 --replacements-per-step 0 > $OUT/frames250.json 2> $OUT/frames250.err
eval timeout 400 python $R/bench.py $COMMON --image 3008 --capture-budget-gb 0 --replacements-per-step 0 --pretrain 600 > $OUT/image3008.json 2> $OUT/image3008.err
for f in partitioning_none segment100_frames100 frames250 image3008; do echo "== $f"; tail -c 400 $OUT/$f.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/$f.json"))
    print(d["config"]["workload"]); print("value", d["value"], "ms/step", d["ms_per_step"], "spr_post", d["samples_per_ray_post"], "setup_s", d["setup_s"])
    for k in d["roofline_kernels"]: print("  ", k["kernel"][:34], k["frac"], k["avg_launch_ms"])
except Exception as e: print("no line:", e)
PY
done
