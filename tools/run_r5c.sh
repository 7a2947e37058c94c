#!/bin/bash
# round 5, call C: what bounds the gather kernels? (a 22 % cut of the level body's vector instructions left their time where it
# was, tools/run_r5a.sh.) Ablations of the level body (no gathers / gathers only / gathers + ds_bpermute), a fifth wavefront per
# SIMD, corner weights formed late; and 16 tile queues side by side in the accumulate kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5c.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default latew w5 abl1 abl2 abl3; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  for mode in march fwd; do
    echo "== lib=$tag mode=$mode" >> $L
    KB_LIB=$lib KB_ONLY=$mode timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:" >> $L
  done
done
trace() {  # per-kernel times of the binned scatter from a kernel trace
  rm -rf /tmp/kt
  KB_LIB=$2 KB_QMAX=$3 KB_REPS=5 KB_ONLY=scatterprof timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/kbench.py > $OUT/kb_scatter_$1.log 2>&1
  grep -E "ms$|records|batch" $OUT/kb_scatter_$1.log >> $L
  f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
  python - >> $L <<PY
import csv, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "scatter" in k or "bwd_tables" in k:
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-40s n=%d last6 (ms): %s" % (k[:40], len(v), " ".join("%.3f" % x for x in v[-6:])))
PY
}
for v in default:: accu16:tools/_build/libhrf_hip_accu16.so:64; do
  IFS=: read tag lib qm <<< "$v"
  echo "== scatter variant $tag" >> $L
  trace $tag "$lib" "${qm:-64}"
done
cat $L | cut -c1-200 | grep -v amdgpu.ids
