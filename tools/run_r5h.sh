#!/bin/bash
# round 5, call H: FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/fetch_calib.hip); scatter kernel
# trace (accumulate: slot-major, finest level first); the HostCapture test; fp32 vs fp16 gradient boundaries at emb 2.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5h
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o m -- tools/microbench/_build/fetch_calib > $OUT/calib_$c.log 2>&1
  f=$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1)
  python - >> $L <<PY
import csv, collections, re
known = {}
for line in open("$OUT/calib_$c.log"):
    m = re.match(r"CALIB (\w+) (\d+)", line)
    if m: known[m.group(1)] = float(m.group(2))
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r["Counter_Name"] == "$c":
        by[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print("# $c (KB as reported x 1024) against the bytes each kernel is known to move; second of two launches")
for k, v in by.items():
    if k in known:
        print("%-20s reported %.4g B   known %.4g B   reported / known %.3f" % (k, v[-1] * 1024, known[k], v[-1] * 1024 / known[k]))
PY
done
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5h.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
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
for mode in mlpbwd march; do
  echo "== mode=$mode" >> $L
  KB_ONLY=$mode timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:" >> $L
done
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_scatter.py tests/test_gpu_compat_tcnn.py tests/test_gpu_round4.py tests/test_gpu_data_parallel.py -q -m gpu >> $L 2>&1
echo "== gradient boundaries at camera_embedding_dim 2 (2 080 steps, identical seeds): fp32 / fp16 / fp16 in the MLP backward only / fp16 in the table scatter only" >> $L
STEPS=2080 timeout 900 python tools/psnr_variance.py default:6 fp16b:6 fp16bmlp:3 fp16btab:3 2>&1 | grep -E "run|segments" | tee $OUT/boundaries_emb2.txt >> $L
cat $L | cut -c1-250 | grep -v amdgpu.ids
