#!/bin/bash
# Collects PMC counters for the gather kernels in separate passes (one rocprofv3 run per counter group).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_prune_march|k_encode4d_fwd" --output-format csv -d /tmp/pmc$i -o p$i -- python $GRAFT_REPO_ROOT/tools/prof_march.py > $OUT/run$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"][:24]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open("$OUT/pass$i.txt","w")
for k,v in agg.items():
    for c,vals in v.items():
        tail=vals[-8:]
        out.write("%s %s n=%d last8_mean=%.6g last8=%s\n"%(k,c,len(vals),sum(tail)/len(tail)," ".join("%.4g"%x for x in tail)))
out.close()
PY
  tail -2 $OUT/run$i.log
done
cat $OUT/pass*.txt
