#!/bin/bash
# round 5, call D: accumulate kernel with the next record window in flight (SB_ACC_PIPE), MLP backward with its weight
# fragments read from LDS per tile instead of parked in registers (MLPB_IL2).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5d
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5d.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default mlpnp; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  echo "== lib=$tag mode=mlpbwd" >> $L
  KB_LIB=$lib KB_ONLY=mlpbwd timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$" >> $L
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
for v in default:: accpipe:tools/_build/libhrf_hip_accpipe.so:64; do
  IFS=: read tag lib qm <<< "$v"
  echo "== scatter variant $tag" >> $L
  trace $tag "$lib" "${qm:-64}"
done
HRF_TEST_LIB=tools/_build/libhrf_hip_accpipe.so timeout 300 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu -k "reproducible or oracle or equals" >> $L 2>&1
echo "pytest accpipe rc=$?" >> $L
cat $L | cut -c1-200 | grep -v amdgpu.ids
