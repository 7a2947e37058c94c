#!/bin/bash
# One entry point for the GPU-side measurements of this repository (round 6 on; the run_r4*.sh / run_r5*.sh one-offs of earlier rounds
# are described in profiles/HISTORY.md). usage, on the GPU box:  bash tools/measure.sh <experiment> [args]
#   gpurun --timeout 900 -- 'bash tools/measure.sh phase'
# Every experiment writes under gpurun_out/<experiment>/ (merged back by gpurun); summaries worth keeping are copied to profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
EXP=${1:-help}; shift
OUT=$R/gpurun_out/$EXP
mkdir -p $OUT
cd $R
L=$OUT/log.txt

warm_cache() {   # one warm-up training shared by every kbench run of the call: parameters + batch in $KB_CACHE
  export KB_WARM=${KB_WARM:-1500} KB_REPS=${KB_REPS:-10} KB_CACHE=/tmp/kb_$EXP.pt
  [ -f $KB_CACHE ] || KB_ONLY=none timeout 600 python tools/kbench.py > $OUT/kb_warm.log 2>&1
}

pmc_kernel() {   # pmc_kernel <lib or ""> <KB_ONLY> <kernel regex> <counters...> : mean of the last 4 launches per counter
  local lib=$1 only=$2 rx=$3; shift 3
  rm -rf /tmp/pk
  KB_LIB=$lib KB_ONLY=$only timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$rx" --output-format csv -d /tmp/pk -o p \
      -- python tools/kbench.py > $OUT/pmc_last.log 2>&1
  local f=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, collections, sys
by = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        by[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in by.items():
        for c, v in d.items():
            tail = v[-4:]
            print("  %-40s %-16s mean_last4 %.6g" % (k, c, sum(tail) / len(tail)))
except Exception as e:
    print("  pmc pass failed:", e)
PY
}

case $EXP in
phase)
  # (i) the no-miss bound of the gather kernels (every table offset folded into a 16 KB window per level table:
  # -DENC_FOLD_MASK, wrong values), (ii) the level order following the chip clock (-DENC_PHASE=1|2|3, bit-identical values) at several
  # clock shifts, with the L2 hit / miss counters of the march for each.
  : > $L
  warm_cache
  for tag in default fold ph1 ph2 ph3; do
    lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
    [ -z "$lib" ] || [ -f "$lib" ] || continue
    sh=""; case $tag in ph*) sh=${SHIFTS:-8,9,10,11,12};; esac   # (ph* libraries: make variant TAG=ph1 EXTRA="-DENC_PHASE=1 -DENC_PHASE_TUNE")
    echo "== lib=$tag" >> $L
    KB_LIB=$lib KB_ONLY=gather KB_SHIFTS=$sh timeout 300 python tools/kbench.py 2>&1 | grep -E "ms$|^march:|^batch:" >> $L
  done
  for tag in default fold ph1 ph2 ph3; do
    lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
    [ -z "$lib" ] || [ -f "$lib" ] || continue
    for s in ${PMC_SHIFTS:-9 10 11}; do
      case $tag in default|fold) [ $s = 10 ] || continue;; esac
      echo "== TCC counters lib=$tag shift=$s" >> $L
      HRF_PHASE_SHIFT=$s HRF_PHASE_SHIFT_FWD=$s pmc_kernel "$lib" gather "k_prune_march|k_encode4d_fwd" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum >> $L
    done
  done
  ;;
pairbench)
  # TA / TCP cost of fetching a cell's x-neighbour corner pair with one 8- or 16-byte load (tools/microbench/pair_bench.hip)
  make -C tools/microbench _build/pair_bench > $OUT/build.log 2>&1
  tools/microbench/_build/pair_bench > $L 2>&1
  ;;
dp)
  # The data-parallel step on one GPU box: (1) the plain `python bench.py --gpus 2` command (self-launch, gloo, both ranks on cuda:0),
  # (2) a one-rank RCCL group with every collective of the step forced (--force-collectives), segment groups 4 and 1.
  : > $L
  SHORT="--pretrain ${PRETRAIN:-400} --trials 1 --steps 20 --warmup 5 --no-cpu-baseline --no-validation --curve ''"
  echo "== plain command, 2 ranks (gloo, same device)" >> $L
  eval timeout 900 python bench.py --gpus 2 --backend gloo --same-device $SHORT > $OUT/dp2_gloo.json 2> $OUT/dp2_gloo.err
  echo "rc=$? lines=$(wc -l < $OUT/dp2_gloo.json)" >> $L
  for g in 4 1 7; do
    echo "== RCCL, one rank, forced collectives, --exchange-groups $g" >> $L
    eval timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
        bench.py --gpus 1 --force-collectives --exchange-groups $g $SHORT > $OUT/rccl1_g$g.json 2> $OUT/rccl1_g$g.err
    echo "rc=$?" >> $L
  done
  python - $OUT >> $L <<'PY'
import json, sys, os
for f in sorted(os.listdir(sys.argv[1])):
    if f.endswith(".json"):
        try:
            d = json.loads(open(os.path.join(sys.argv[1], f)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "no line:", e); continue
        print(f, "value %.0f ms/step %.3f n_gpus %d exchange issued %s exposed %s bytes %s groups %s" % (
            d["value"], d["ms_per_step"], d["n_gpus"], d.get("gradient_exchange_ms_per_step"), d.get("gradient_exchange_exposed_ms_per_step"),
            d.get("gradient_exchange_bytes_per_rank_last_step"), d.get("gradient_exchange_groups")))
        print("   issue order:", d.get("gradient_exchange_issue_order_last_step"))
        print("   collectives:", d.get("collectives"))
        print("   kernels:", d.get("kernel_ms_per_step"))
PY
  ;;
*)
  echo "experiments: phase pairbench dp"; exit 1;;
esac
echo "done: $OUT"
