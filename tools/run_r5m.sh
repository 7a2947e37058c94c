#!/bin/bash
# round 5, call M: smoke(), the N = 2 bench path end to end on one GPU (gloo, functional only), tools/run_scale.sh on the one GPU of the box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5m
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 \
  --backend gloo --same-device --trials 2 --pretrain 300 --steps 10 --warmup 5 --no-cpu-baseline --curve '' --validation-views 2 > $OUT/dp2_gloo.json 2> $OUT/dp2_gloo.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/dp2_gloo.json").read().strip().splitlines()[-1])
    print("dp2 gloo same-device: value %.0f ms %.3f n_gpus %d" % (d["value"], d["ms_per_step"], d["n_gpus"]), d["collectives"], d["gradient_exchange_ms_per_step"], [t["value"] for t in d["trials"]])
except Exception as e:
    print("no line:", e); print(open("$OUT/dp2_gloo.err").read()[-2500:])
PY
NS="1 2" SCALINGS="weak" EXCHANGES="sharded" EXTRA="--pretrain 300" timeout 600 bash tools/run_scale.sh $OUT/scale | tail -5
