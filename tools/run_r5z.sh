#!/bin/bash
# round 5, call Z: ablations of the emit kernel (measurement-only builds, WRONG results): -DSB_ABLATE=1 spreads the LDS slot counters over
# eight replicas (same-address conflicts / 8), -DSB_ABLATE=2 writes no records. Per-kernel times from a kernel trace of KB_ONLY=scatterprof.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5z
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=10 KB_CACHE=/tmp/kb_r5z.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in ${TAGS:-default emabl1 emabl2}; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  echo "== lib=$tag" >> $L
  rm -rf /tmp/kt
  KB_LIB=$lib KB_ONLY=scatterprof timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/kbench.py > $OUT/kb_$tag.log 2>&1
  f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
  python - >> $L <<PY
import csv, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "scatter_emit" in k or "scatter_accumulate" in k:
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-40s n=%d last6 (ms): %s" % (k[:40], len(v), " ".join("%.3f" % x for x in v[-6:])))
PY
done
