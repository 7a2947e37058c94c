#!/bin/bash
# pipelined pieces: equivalence test + bench at 1 / 2 / 4 pieces. usage: bash tools/run_pieces.sh TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pieces}
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_ref_fixtures.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
COMMON="--no-cpu-baseline --curve '' --no-validation --pretrain 2000 --steps 100"
for n in 4 1 2; do
  HRF_PIECES=$n eval timeout 200 python bench.py $COMMON > $OUT/p$n.json 2> $OUT/p$n.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/p$n.json"))
    print("pieces $n: value", d["value"], "ms/step", d["ms_per_step"], "spr", d["samples_per_ray_post"], "psnr", d["train_psnr_db"])
    for k in d["roofline_kernels"]: print("  ", k["kernel"][:40], k["frac"], k["avg_launch_ms"])
except Exception as e: print("no line:", e); print(open("$OUT/p$n.err").read()[-1500:])
PY
done
