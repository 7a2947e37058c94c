#!/bin/bash
# MFMA utilisation of the MLP kernels (and of the fused march) over a short training run; counters in their own passes.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_mfma}
mkdir -p $OUT
for c in MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES; do
  PM_WARM=400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_mlp_bwd|k_color_fwd|k_density_fwd|k_prune_march" --output-format csv -d /tmp/mf_$c -o m -- python $GRAFT_REPO_ROOT/tools/prof_mlp.py > $OUT/run_$c.log 2>&1
  f=$(find /tmp/mf_$c -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$f")) if r["Counter_Name"]=="$c"]
by=collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
with open("$OUT/$c.txt","w") as o:
    for k,v in by.items():
        tail=v[-6:]
        line="%-42s $c n=%d mean_last6=%.6g  last6=%s"%(k,len(v),sum(tail)/len(tail)," ".join("%.4g"%x for x in tail))
        o.write(line+"\n"); print(line)
PY
done
