#!/bin/bash
# GPU tests + bench lines fp16 / bf16 with the device-side GradScaler. usage: bash tools/run_scaler.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-scaler}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
COMMON="--no-cpu-baseline --curve '' --validation-views 2 --pretrain 2000 --steps 60"
eval timeout 300 python bench.py $COMMON --mlp-precision fp16 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err
eval timeout 300 python bench.py $COMMON --mlp-precision bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
for f in fp16 bf16; do echo "== $f"; tail -c 300 $OUT/bench_$f.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$f.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "spr_post", d["samples_per_ray_post"], "psnr", d.get("train_psnr_db"), "val", d.get("validation", {}).get("psnr_db_mean"))
    print(d["grad_scaler"]["skipped_steps_total"], d["grad_scaler"]["scale_now"])
    for k in d["roofline_kernels"]: print("  ", k["kernel"][:40], k["frac"], k["avg_launch_ms"])
    print(d["regime_curve"])
except Exception as e: print("no line:", e)
PY
done
