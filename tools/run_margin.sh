#!/bin/bash
# speculation margin sweep. usage: bash tools/run_margin.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-margin}
mkdir -p $OUT
cd $R
COMMON="--no-cpu-baseline --curve '' --no-validation --pretrain 2000 --steps 100"
for cfg in "1.03 1" "1.08 1" "1.05 3" "1.12 1"; do
  set -- $cfg
  HRF_SPEC_MARGIN=$1 HRF_SPEC_HISTORY=$2 eval timeout 200 python bench.py $COMMON > $OUT/m_$1_$2.json 2> $OUT/m_$1_$2.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/m_$1_$2.json"))
    print("margin $1 hist $2: value", d["value"], "ms/step", d["ms_per_step"], "launches/step", d["prune_march_launches_per_step"], "marched/used", d["drawn_rays_marched_over_used"], "spr", d["samples_per_ray_post"])
except Exception as e: print("no line:", e)
PY
done
