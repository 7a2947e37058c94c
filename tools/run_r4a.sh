#!/bin/bash
# round-4 check A: GPU test suite (with the per-entry update diagnostics), then bench lines of the default configuration with the
# step variants, and of the one-segment (2^18 tables) configuration. usage: bash tools/run_r4a.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4a}
mkdir -p $OUT
cd $R
export HRF_TEST_DIAG=$OUT/diag.txt
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
COMMON="--no-cpu-baseline --curve '' --no-validation --steps 60"
run() {  # name, args
  eval timeout 240 python bench.py $COMMON $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json"))
    print("$1: value %.0f ms/step %.3f spr %.2f psnr %.2f" % (d["value"], d["ms_per_step"], d["samples_per_ray_post"], d["train_psnr_db"]), d["kernel_ms_per_step"])
except Exception as e:
    print("$1: no line:", e); print(open("$OUT/$1.err").read()[-1500:])
PY
}
run default "--pretrain 2000 --kernel-breakdown"
run split "--pretrain 2000 --mlp-backward split --kernel-breakdown"
run serialvec "--pretrain 2000 --no-overlap-vectors --kernel-breakdown"
run none "--pretrain 2000 --partitioning none --kernel-breakdown"
run none_atomic "--pretrain 2000 --partitioning none --table-scatter atomic --kernel-breakdown"
