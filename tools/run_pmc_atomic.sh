#!/bin/bash
# L2 atomic requests of the gradient scatter kernels per launch (TCC_ATOMIC_sum), against the measured 21 G requests/s.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_atomic
mkdir -p $OUT
for c in TCC_ATOMIC_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum; do
  PM_WARM=400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_encode4d_bwd" --output-format csv -d /tmp/at_$c -o m -- python $GRAFT_REPO_ROOT/tools/prof_mlp.py > $OUT/run_$c.log 2>&1
  f=$(find /tmp/at_$c -name "*counter_collection.csv" | head -1)
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
  tail -1 $OUT/run_$c.log
done
