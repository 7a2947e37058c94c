#!/bin/bash
# round-4 check B: GPU test suite; kbench A/B of the gather kernels with / without v_fma_mix (one warm-up training, KB_CACHE);
# per-kernel times of the binned scatter on the one-segment (2^18 tables) model; SQ counters of k_mlp_bwd.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4b}
mkdir -p $OUT
cd $R
export HRF_TEST_DIAG=$OUT/diag.txt
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log | cut -c1-220
# ---- kbench: default model, 2000 warm-up steps
export KB_WARM=2000 KB_REPS=20 KB_CACHE=/tmp/kb_r4b.pt
rm -f $KB_CACHE
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for lib in "" tools/_build/libhrf_hip_nomix.so; do
  for mode in march fwd; do
    echo "== lib=${lib:-default} mode=$mode"
    KB_LIB=$lib KB_ONLY=$mode timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:"
  done
done
echo "== scatter (default model)"
KB_ONLY=scatterprof timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|records"
echo "== mlp_bwd"
KB_ONLY=mlpbwd timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$"
# SQ counters of k_mlp_bwd (3 SQ groups + TCC)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/sqm$i
  KB_ONLY=mlpbwd timeout 100 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_mlp_bwd" --output-format csv -d /tmp/sqm$i -o p -- python $R/tools/kbench.py > $OUT/run_mlpbwd_$i.log 2>&1
  f=$(find /tmp/sqm$i -name "*counter_collection.csv" | head -1)
  python - <<PY >> $OUT/sq_k_mlp_bwd.txt
import csv, collections
by = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open("$f")):
        by[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in by.items():
        for c, v in d.items():
            tail = v[-4:]
            print("%-42s %-36s n=%d mean_last4 %.6g" % (k, c, len(v), sum(tail) / len(tail)))
except Exception as e:
    print("# pass $i failed:", e)
PY
done
cat $OUT/sq_k_mlp_bwd.txt | cut -c1-140
# ---- one-segment model (2^18-entry tables): per-kernel times of the binned scatter from a kernel trace
export KB_SEGMENTS=50 KB_CACHE=/tmp/kb_r4b_none.pt KB_REPS=5
rm -f $KB_CACHE
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm_none.log 2>&1
rm -rf /tmp/kt
KB_ONLY=scatterprof timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/kbench.py > $OUT/kb_none_scatter.log 2>&1
grep -E "ms$|records|batch" $OUT/kb_none_scatter.log
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "scatter" in k or "bwd_tables" in k:
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-40s n=%d last6 (ms): %s" % (k[:40], len(v), " ".join("%.3f" % x for x in v[-6:])))
PY
KB_ONLY=march timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:"
