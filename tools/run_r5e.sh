#!/bin/bash
# round 5, call E: the whole GPU suite on the current build; march timing against the round-4 library (the level loop gained a
# bound for models with fewer levels); the accumulate kernel's finest-level-first dispatch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5e
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log | cut -c1-250 >> $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5e.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default base; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  for mode in march fwd mlpbwd; do
    echo "== lib=$tag mode=$mode" >> $L
    KB_LIB=$lib KB_ONLY=$mode timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:" >> $L
  done
done
rm -rf /tmp/kt
KB_REPS=5 KB_ONLY=scatterprof timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/kbench.py > $OUT/kb_scatter.log 2>&1
grep -E "ms$|records|batch" $OUT/kb_scatter.log >> $L
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
cat $L | cut -c1-220 | grep -v amdgpu.ids
