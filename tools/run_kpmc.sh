#!/bin/bash
# PMC passes over the binned-scatter kernels of tools/kbench.py (KB_ONLY=scatterprof): one rocprofv3 run per counter group
# (SQ: 8 slots per pass). Last launches only (the micro-benchmark's, after the warm-up training). -> gpurun_out/$1/
# KP_N=3 stops after the trace and the two SQ passes. KP_ONLY / KP_REGEX select another kbench mode and kernel, e.g.
# KP_ONLY=fwd KP_REGEX=k_encode4d_fwd, KP_ONLY=mlpbwd KP_REGEX=k_mlp_bwd (default: the binned scatter).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-kpmc}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export KB_ONLY=${KP_ONLY:-scatterprof}
REGEX=${KP_REGEX:-k_scatter}
# one warm-up training for all passes: the first (unprofiled) run leaves parameters + batch in KB_CACHE, the passes start from it
export KB_CACHE=/tmp/kb_cache_$TAG.pt
rm -f $KB_CACHE
python $R/tools/kbench.py > $OUT/run0.log 2>&1
grep -E "batch:|records per sample|scatter" $OUT/run0.log
i=0
for grp in "TRACE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$KP_N" ] && [ $i -gt $KP_N ]; then break; fi
  rm -rf /tmp/kp$i
  if [ "$grp" = "TRACE" ]; then
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kp$i -o p -- python $R/tools/kbench.py > $OUT/run$i.log 2>&1
    f=$(find /tmp/kp$i -name "*kernel_trace.csv" | head -1)
    python - <<PY > $OUT/last_launches.txt
import csv, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "scatter" in k or "bwd_tables" in k or "$REGEX" in k:
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-40s n=%d last6 (ms): %s" % (k[:40], len(v), " ".join("%.3f" % x for x in v[-6:])))
PY
    cat $OUT/last_launches.txt; grep -E "records per sample|scatter" $OUT/run$i.log
    continue
  fi
  rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$REGEX" --output-format csv -d /tmp/kp$i -o p -- python $R/tools/kbench.py > $OUT/run$i.log 2>&1
  f=$(find /tmp/kp$i -name "*counter_collection.csv" | head -1)
  python - <<PY | tee -a $OUT/counters.txt
import csv, collections
by = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open("$f")):
        by[r["Kernel_Name"].split("(")[0][:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in by.items():
        for c, v in d.items():
            tail = v[-4:]
            print("%-34s %-24s mean_last4 %.6g" % (k, c, sum(tail) / len(tail)))
except Exception as e:
    print("pass $i failed:", e)
PY
  tail -2 $OUT/run$i.log | cut -c1-200
done
