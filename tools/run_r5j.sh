#!/bin/bash
# round 5, call J: configs[2] (1x, 3008^2) and configs[4] (1000 frames) shapes on one GPU with a LIVE replacer streaming from the pinned
# host capture; the data-parallel step on a one-rank RCCL group (self-check + exchange timing).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5j
mkdir -p $OUT
cd $R
COMMON="--trials 1 --no-cpu-baseline --curve '' --validation-views 4"
eval timeout 700 python bench.py --image 3008 --host-capture-gb ${HOST_GB:-24} --pretrain 1000 $COMMON > $OUT/config_image3008.json 2> $OUT/config_image3008.err
eval timeout 900 python bench.py --frames 1000 --host-capture-gb ${HOST_GB:-24} --pretrain 1000 --mlp-precision bf16 $COMMON > $OUT/config_frames1000.json 2> $OUT/config_frames1000.err
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 \
  --force-collectives --backend nccl --pretrain 500 --steps 40 --trials 1 --no-cpu-baseline --no-validation --curve '' > $OUT/rccl_world1.json 2> $OUT/rccl_world1.err
python - <<PY
import json
for n in ("config_image3008", "config_frames1000", "rccl_world1"):
    try:
        d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print(n, "value %.0f ms/step %.3f spr %.2f psnr %s" % (d["value"], d["ms_per_step"], d["samples_per_ray_post"], d.get("validation_psnr_db")),
              d["replacer"], d.get("gradient_exchange_ms_per_step"), d.get("collectives"), "setup_s", d["setup_s"])
        print("   ", d["config"]["workload"][:160])
    except Exception as e:
        print(n, "no line:", e); print(open("$OUT/%s.err" % n).read()[-1500:])
PY
# SQ counters of the march on the rewritten level body (lane-instructions per encoded sample)
SQ_ONLY="march:k_prune_march" KB_WARM=2000 timeout 600 bash tools/run_sq_r04.sh r5j_sq 2>&1 | tail -40
