#!/bin/bash
# tools/run_scale.sh [OUT_DIR] -- the 1 / 2 / 4 / 8-GPU sweep of bench.py on ONE node, one JSON line per point:
#   N in {1, 2, 4, 8} x scaling in {weak, strong} x exchange in {sharded, allreduce}   (N = 1 runs once per scaling: no exchange)
# -> OUT_DIR/scale_n<N>_<scaling>_<exchange>.json (+ .err), and OUT_DIR/summary.txt (value, ms / step, exchange ms / step per point).
# Every rank is one process on one GPU (torch.distributed.run, backend nccl = RCCL over xGMI); at start-up each rank's
# TableShardExchange.self_check() compares the in-place reduce_scatter_tensor + all_gather_into_tensor it trains with against
# all_reduce on a 1 MB probe and raises on a mismatch. Nothing is estimated here: a point that cannot run (fewer GPUs than N)
# is recorded as skipped. Environment: NS="1 2 4 8" SCALINGS="weak strong" EXCHANGES="sharded allreduce" STEPS=20 WARMUP=5
# TRIALS=1 PORT=29600 EXTRA="...bench.py flags...".
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/scale}
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPUS=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
PORT=${PORT:-29600}
: > "$OUT/summary.txt"
for N in ${NS:-1 2 4 8}; do
  for SC in ${SCALINGS:-weak strong}; do
    for EX in ${EXCHANGES:-sharded allreduce}; do
      [ "$N" = 1 ] && [ "$EX" != "sharded" ] && continue
      TAG=scale_n${N}_${SC}_${EX}
      if [ "$N" -gt "$GPUS" ]; then
        echo "{\"skipped\": \"$N GPUs asked, $GPUS visible\", \"n_gpus\": $N, \"scaling\": \"$SC\", \"exchange\": \"$EX\"}" > "$OUT/$TAG.json"
        echo "$TAG skipped ($GPUS GPUs visible)" >> "$OUT/summary.txt"
        continue
      fi
      PORT=$((PORT + 1))
      ARGS="--gpus $N --steps ${STEPS:-20} --warmup ${WARMUP:-5} --trials ${TRIALS:-1} --scaling $SC --exchange $EX --no-cpu-baseline --no-validation --curve '' ${EXTRA:-}"
      if [ "$N" = 1 ]; then
        eval timeout ${TIMEOUT:-900} python bench.py $ARGS > "$OUT/$TAG.json" 2> "$OUT/$TAG.err"
      else
        eval timeout ${TIMEOUT:-900} python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
          bench.py $ARGS > "$OUT/$TAG.json" 2> "$OUT/$TAG.err"
      fi
      python - "$OUT/$TAG.json" "$TAG" >> "$OUT/summary.txt" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-32s value %12.0f rays/s  %7.3f ms/step  samples/ray %5.2f  exchange %s ms/step  %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["samples_per_ray_post"], d.get("gradient_exchange_ms_per_step"),
        ",".join(d.get("collectives", {}).get("calls", []))[:120]))
except Exception as e:
    print("%-32s no line (%s): see the .err file" % (sys.argv[2], e))
PY
    done
  done
done
cat "$OUT/summary.txt"
