#!/bin/bash
# Bench lines of the other single-GPU configurations in round 3: configs[3] shape (250 frames, 32 segments), configs[4] shape
# (1 000 frames: 125+ segments, images rendered on demand), configs[2] (1x scale, 3008^2), the bf16 MLP variant, and a
# single 2^18 segment (level tables above 65 536 entries: the atomic scatter). usage: bash tools/run_configs_r03.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-configs}
mkdir -p $OUT
COMMON="--no-cpu-baseline --curve '' --validation-views 1 --pretrain 1000 --steps 40"
eval timeout 400 python $R/bench.py $COMMON --frames 250 --capture-budget-gb 0 --replacements-per-step 0 > $OUT/frames250.json 2> $OUT/frames250.err
eval timeout 600 python $R/bench.py $COMMON --frames 1000 --capture-budget-gb 0 --replacements-per-step 0 > $OUT/frames1000.json 2> $OUT/frames1000.err
eval timeout 400 python $R/bench.py $COMMON --image 3008 --capture-budget-gb 0 --replacements-per-step 0 --pretrain 600 > $OUT/image3008.json 2> $OUT/image3008.err
eval timeout 300 python $R/bench.py --no-cpu-baseline --curve "''" --validation-views 2 --mlp-precision bf16 > $OUT/bf16_mlp.json 2> $OUT/bf16_mlp.err
eval timeout 300 python $R/bench.py $COMMON --partitioning none > $OUT/partitioning_none.json 2> $OUT/partitioning_none.err
for f in frames250 frames1000 image3008 bf16_mlp partitioning_none; do echo "== $f"; tail -c 300 $OUT/$f.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print(d["config"]["workload"][:160]); print("value", d["value"], "ms/step", d["ms_per_step"], "spr_post", d["samples_per_ray_post"], "psnr", d["train_psnr_db"], "val", d.get("validation_psnr_db"), "setup_s", d["setup_s"])
    for k in d["roofline_kernels"]: print("  ", k["kernel"][:44], k["frac"], k["avg_launch_ms"])
except Exception as e: print("no line:", e)
PY
done
