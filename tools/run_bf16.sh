#!/bin/bash
# GPU tests + fp16 / bf16 A/B of the bench line (same settings). usage: bash tools/run_bf16.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-bf16}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
COMMON="--no-cpu-baseline --curve '' --validation-views 2 --pretrain 2000 --steps 60 --kernel-breakdown"
eval timeout 300 python bench.py $COMMON --mlp-precision bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
eval timeout 300 python bench.py $COMMON --mlp-precision fp16 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err
for f in bf16 fp16; do echo "== $f"; tail -c 300 $OUT/bench_$f.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$f.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "spr_post", d["samples_per_ray_post"], "psnr", d.get("train_psnr_db"), "val", d.get("validation"))
    print({k: v for k, v in d.get("kernel_ms_per_step", d.get("breakdown", {})).items()} if isinstance(d.get("kernel_ms_per_step", d.get("breakdown")), dict) else "")
except Exception as e: print("no line:", e)
PY
done
