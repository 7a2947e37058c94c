#!/bin/bash
# round 5, call B: x-neighbour pair loads in the level body (ENC_PAIR=8) and scatter variants (half staging -> 3 emit workgroups
# per CU; 4096-entry accumulate chunks -> 2 accumulate workgroups per CU), each timed on one cached batch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5b
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5b.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
HRF_TEST_LIB=tools/_build/libhrf_hip_pair8.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_level_body or encode4d_forward or fused_prune_march" >> $L 2>&1
echo "pytest pair8 rc=$?" >> $L
for lib in "" tools/_build/libhrf_hip_pair8.so; do
  for mode in march fwd; do
    echo "== lib=${lib:-default} mode=$mode" >> $L
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
for v in default:: halfg0:tools/_build/libhrf_hip_halfg0.so:64 acc12:tools/_build/libhrf_hip_acc12.so:128 acc12t512:tools/_build/libhrf_hip_acc12t512.so:128; do
  IFS=: read tag lib qm <<< "$v"
  echo "== scatter variant $tag" >> $L
  trace $tag "$lib" "${qm:-64}"
done
HRF_TEST_LIB=tools/_build/libhrf_hip_acc12.so timeout 600 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu >> $L 2>&1
echo "pytest acc12 scatter rc=$?" >> $L
timeout 600 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu >> $L 2>&1
echo "pytest default scatter rc=$?" >> $L
cat $L | cut -c1-200
