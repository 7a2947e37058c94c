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
pair)
  # the level body with paired x-neighbour fetches (-DENC_PAIR=1, bit-identical) against the shipped one: kbench gather kernels, the
  # march's L2 request counters, then the parity suite on the variant library
  : > $L
  warm_cache
  for tag in ${TAGS:-default pair default pair}; do
    lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
    echo "== lib=$tag" >> $L
    KB_LIB=$lib KB_ONLY=gather timeout 300 python tools/kbench.py 2>&1 | grep -E "ms$|^march:|^batch:" >> $L
  done
  for tag in default pair; do
    lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
    echo "== TCC counters lib=$tag" >> $L
    pmc_kernel "$lib" gather "k_prune_march|k_encode4d_fwd" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum >> $L
  done
  HRF_TEST_LIB=tools/_build/libhrf_hip_pair.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ref_fixtures.py tests/test_gpu_compat_tcnn.py -x -q -m gpu 2>&1 | tail -4 >> $L
  ;;
pmc)
  # HBM-side traffic (FETCH_SIZE, WRITE_SIZE) and memory-side atomic requests (TCC_ATOMIC_sum) of the gather / scatter kernels per
  # unit of work, and MfmaUtil of the MLP kernels: each counter in its own rocprofv3 pass (kernel trace + pmc only) over
  # tools/pmc_driver.py = the bench configuration ("$@": bench.py flags) after PM_WARM training steps. -> FETCH_SIZE.txt ...,
  # traffic.json, mfma.json with the fingerprint of the kernel sources (copy to profiles/r06_traffic.json, profiles/r06_mfma.json:
  # bench.py reports them when the fingerprint matches the sources it runs).
  export PM_WARM=${PM_WARM:-1500}
  : > $L
  KERN="k_prune_march|k_encode4d_fwd|k_scatter_emit|k_scatter_accumulate|k_encode4d_bwd_tables_lm|k_encode4d_bwd_vectors"
  for c in FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum; do
    rm -rf /tmp/pm_$c
    rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$KERN" --output-format csv -d /tmp/pm_$c -o m -- python tools/pmc_driver.py "$@" > $OUT/run_$c.log 2>&1
    python tools/pmc_summary.py counter "$(find /tmp/pm_$c -name '*counter_collection.csv' | head -1)" $OUT/run_$c.log $c $OUT >> $L 2>&1
  done
  python tools/pmc_summary.py traffic $OUT >> $L 2>&1
  rm -rf /tmp/mf_u
  rocprofv3 --kernel-trace --pmc MfmaUtil --kernel-include-regex "k_mlp_bwd|k_color_fwd|k_density_fwd|k_prune_march" \
      --output-format csv -d /tmp/mf_u -o m -- python tools/pmc_driver.py "$@" > $OUT/run_MfmaUtil.log 2>&1
  python tools/pmc_summary.py mfma "$(find /tmp/mf_u -name '*counter_collection.csv' | head -1)" $OUT >> $L 2>&1
  ;;
profile)
  # rocprofv3 kernel trace + stats of one trial of the bench (the timed region = the last 60 steps) -> kernel_stats.csv,
  # timed_region.txt (tools/gaps.py: window / busy / idle per step, per-kernel time, the largest gaps), the bench's own line
  rm -rf /tmp/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python bench.py --trials 1 --steps 60 --warmup 20 \
      --no-cpu-baseline --no-validation --no-other-configs --kernel-window 0 --curve '' "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
  cp "$(find /tmp/prof -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
  python tools/gaps.py "$(find /tmp/prof -name '*kernel_trace.csv' | head -1)" 60 > $OUT/timed_region.txt 2>&1
  head -120 $OUT/timed_region.txt | cut -c1-150
  ;;
kpmc)
  # SQ / TCC counter passes over one kernel of tools/kbench.py: bash tools/measure.sh kpmc <KB_ONLY mode> <kernel regex> [lib]
  : > $L
  warm_cache
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    echo "== $grp" >> $L
    pmc_kernel "${3:-}" "${1:-scatterprof}" "${2:-k_scatter}" $grp >> $L
  done
  ;;
scale)
  # the 1 / 2 / 4 / 8-GPU sweep of bench.py on ONE node, one JSON line per point (the plain `python bench.py --gpus N` form: bench.py
  # starts its own ranks): N x scaling x exchange -> scale_n<N>_<scaling>_<exchange>.json + summary.txt. A point that cannot run
  # (fewer GPUs than N) is recorded as skipped; nothing is estimated. NS="1 2 4 8" SCALINGS="weak strong" EXCHANGES="sharded allreduce"
  GPUS=$(python -c "import torch; print(torch.cuda.device_count())")
  : > $OUT/summary.txt
  for N in ${NS:-1 2 4 8}; do for SC in ${SCALINGS:-weak strong}; do for EX in ${EXCHANGES:-sharded allreduce}; do
    [ "$N" = 1 ] && [ "$EX" != "sharded" ] && continue
    TAG=scale_n${N}_${SC}_${EX}
    if [ "$N" -gt "$GPUS" ]; then
      echo "{\"skipped\": \"$N GPUs asked, $GPUS visible\", \"n_gpus\": $N, \"scaling\": \"$SC\", \"exchange\": \"$EX\"}" > $OUT/$TAG.json
      echo "$TAG skipped ($GPUS GPUs visible)" >> $OUT/summary.txt
      continue
    fi
    eval timeout ${TIMEOUT:-900} python bench.py --gpus $N --steps ${STEPS:-20} --warmup ${WARMUP:-5} --trials ${TRIALS:-1} --scaling $SC \
        --exchange $EX --no-cpu-baseline --no-validation --curve "''" ${EXTRA:-} > $OUT/$TAG.json 2> $OUT/$TAG.err
    python - $OUT/$TAG.json $TAG >> $OUT/summary.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-32s value %12.0f rays/s  %7.3f ms/step  samples/ray %5.2f  exchange issued %s exposed %s ms/step  %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["samples_per_ray_post"], d.get("gradient_exchange_ms_per_step"),
        d.get("gradient_exchange_exposed_ms_per_step"), ",".join(d.get("collectives", {}).get("calls", []))[:120]))
except Exception as e:
    print("%-32s no line (%s): see the .err file" % (sys.argv[2], e))
PY
  done; done; done
  cat $OUT/summary.txt
  ;;
ab)
  # in-process A/B aids of bench.py on ONE trajectory (40-step windows, four rounds each): the scan (one-pass look-back vs the two-launch
  # form), the training loop on a high-priority stream, the vector half of the backward on its own stream
  timeout 1200 python bench.py --trials 1 --steps 20 --warmup 5 --no-cpu-baseline --no-validation --no-other-configs --curve '' \
      --ab-env HRF_SCAN_TWO_PASS --ab-main-priority --ab-overlap-vectors "$@" > $OUT/line.json 2> $L
  grep "^AB\|priority" $L
  ;;
stepbench)
  # ms per step of THIS tree and of the round-5 tree (_r05/: `git archive ea9000b humanrf_amd include bench.py`, built in place) at one
  # frozen model state (tools/stepbench.py): the regime-free form of "did the step get faster"
  : > $L
  CK=/tmp/sb_ck.pt
  timeout 900 python tools/stepbench.py train $CK "$@" >> $L 2>&1
  for rep in 1 2; do
    echo "== this tree (rep $rep)" >> $L
    timeout 600 python tools/stepbench.py measure $CK "$@" 2>&1 | grep "^round" >> $L
    if [ -d _r05 ]; then
      echo "== round-5 tree (rep $rep)" >> $L
      (cd _r05 && timeout 600 python ../tools/stepbench.py measure $CK "$@" 2>&1 | grep "^round\|Error\|error" >> $L)
    fi
  done
  cat $L
  ;;
sweep)
  # engine settings / library variants at one frozen model state (tools/stepbench.py): SWEEP="name|SB_SET|SB_LIB;..." (| separated)
  : > $L
  CK=/tmp/sb_ck.pt
  [ -f $CK ] || timeout 900 python tools/stepbench.py train $CK >> $L 2>&1
  IFS=';' read -ra ITEMS <<< "${SWEEP:-base||}"
  for rep in $(seq 1 ${REPS:-2}); do
    for it in "${ITEMS[@]}"; do
      IFS='|' read -r name set lib <<< "$it"
      echo "== $name (rep $rep)  SB_SET=$set SB_LIB=$lib" >> $L
      SB_SET="$set" SB_LIB="$lib" SB_ROUNDS=${SB_ROUNDS:-4} timeout 600 python tools/stepbench.py measure $CK 2>&1 | grep "^round\|^set\|Error" >> $L
    done
  done
  cat $L
  ;;
abstep)
  # in-process A/B of engine attributes at one frozen model state: ABS="attr=A|B;attr2=A|B" (each in its own process, rounds alternate)
  : > $L
  CK=/tmp/sb_ck.pt
  [ -f $CK ] || timeout 900 python tools/stepbench.py train $CK >> $L 2>&1
  IFS=';' read -ra ITEMS <<< "$ABS"
  for it in "${ITEMS[@]}"; do
    echo "== $it" >> $L
    SB_AB="$it" SB_ROUNDS=${SB_ROUNDS:-10} timeout 900 python tools/stepbench.py measure $CK 2>&1 | grep "^round\|^AB\|Error" >> $L
  done
  grep "^==\|^AB" $L
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
  for g in 4 7; do
    echo "== RCCL, one rank, forced collectives, --exchange-groups $g --no-exchange-signals (one accumulate launch per group)" >> $L
    eval timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
        bench.py --gpus 1 --force-collectives --exchange-groups $g --no-exchange-signals $SHORT > $OUT/rccl1_g${g}_launches.json 2> $OUT/rccl1_g${g}_launches.err
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
            d.get("gradient_exchange_bytes_per_rank_last_step"), d.get("gradient_exchange_groups")), "|", d.get("gradient_exchange_accumulate_mode"))
        print("   issue order:", d.get("gradient_exchange_issue_order_last_step"))
        print("   collectives:", d.get("collectives"))
        print("   kernels:", d.get("kernel_ms_per_step"))
PY
  ;;
sigbench)
  # the accumulate half of the scatter alone: one launch / one per group / one signalled launch (hrf_scatter_accumulate_signalled), the
  # default library and the fence variants (make -C humanrf_amd/csrc variant TAG=sf1 EXTRA=-DSB_SIGNAL_FENCE=1, ...=2); and the probe of
  # hipStreamWaitValue64's latency against a running kernel
  : > $L
  mkdir -p tools/microbench/_build
  hipcc --offload-arch=gfx950 -O2 tools/microbench/wait_value_probe.hip -o tools/microbench/_build/wait_value_probe && \
    timeout 60 tools/microbench/_build/wait_value_probe >> $L 2>&1
  timeout 200 python tools/sigbench.py 2>&1 | grep -v amdgpu.ids >> $L
  for v in 1 2; do
    [ -f tools/_build/libhrf_hip_sf$v.so ] && KB_LIB=tools/_build/libhrf_hip_sf$v.so timeout 200 python tools/sigbench.py 2>&1 | grep -v amdgpu.ids >> $L
  done
  cat $L
  ;;
*)
  echo "experiments: phase pair pairbench dp gradparity pmc profile kpmc scale ab stepbench sweep abstep sigbench"; exit 1;;
esac
echo "done: $OUT"
