#!/bin/bash
# SQ / TA / TCP counters of the gather, scatter and MLP-backward kernels on the headline regime (tools/kbench.py after KB_WARM
# training steps; ONE warm-up training shared by every pass through KB_CACHE). One rocprofv3 run per counter group
# (--kernel-trace + --pmc only). usage: [SQ_ONLY="mlpbwd:k_mlp_bwd"] bash tools/run_sq_r04.sh TAG  -> gpurun_out/TAG/{sq_<kernel>.txt, *.log}
# (The TA_* and TCP_* groups of the first version are gone: on this pool a pass with TA_BUSY_avr / TCP_PENDING_STALL_CYCLES_sum
# aborts or hangs until its timeout -- 5 minutes of box time each, profiles/README.md.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-sq}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export KB_CACHE=/tmp/kb_cache_$TAG.pt
rm -f $KB_CACHE
export KB_WARM=${KB_WARM:-2000}
KB_ONLY=none python $R/tools/kbench.py > $OUT/warm.log 2>&1
tail -3 $OUT/warm.log
FP=$(cd $R && python -c "import bench; print(bench.kernel_source_fingerprint()[:12])")
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES"
         "SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE"
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
for spec in ${SQ_ONLY:-"march:k_prune_march" "fwd:k_encode4d_fwd" "scatterprof:k_scatter_emit|k_scatter_accumulate" "mlpbwd:k_mlp_bwd"}; do
  mode=${spec%%:*}; regex=${spec#*:}
  name=$(echo $regex | tr '|' '_')
  F=$OUT/sq_$name.txt
  echo "# tools/run_sq_r04.sh: rocprofv3 --kernel-trace --pmc <group>, one group per pass, KB_ONLY=$mode tools/kbench.py after $KB_WARM training steps; kernel sources $FP; mean of the last 4 launches" > $F
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    rm -rf /tmp/sq$i
    KB_ONLY=$mode timeout 100 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$regex" --output-format csv -d /tmp/sq$i -o p -- python $R/tools/kbench.py > $OUT/run_${mode}_$i.log 2>&1
    f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1)
    python - <<PY >> $F
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
    print("# pass $i ($grp) failed:", e)
PY
    grep -E "ms$|batch:|march:|records per" $OUT/run_${mode}_$i.log | head -4 >> $OUT/times_$name.txt
  done
  echo "== $name"; cat $F | cut -c1-160
done
