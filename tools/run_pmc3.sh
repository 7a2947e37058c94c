#!/bin/bash
# March HBM-side traffic per encoded sample (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide prescribes).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  PM_WARM=1000 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_prune_march" --output-format csv -d /tmp/m_$c -o m -- python $GRAFT_REPO_ROOT/tools/prof_march.py > $OUT/march_run_$c.log 2>&1
  f=$(find /tmp/m_$c -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, re
rows=[r for r in csv.DictReader(open("$f")) if r["Counter_Name"]=="$c"]
vals=[float(r["Counter_Value"]) for r in rows][-8:]
ev=[int(m.group(1)) for m in re.finditer(r"evaluated (\d+)", open("$OUT/march_run_$c.log").read())][-8:]
per=[v*1024/e for v,e in zip(vals,ev)]
s="last 8 k_prune_march dispatches $c (KB): "+" ".join("%.0f"%v for v in vals)+"\nencoded samples of the same dispatches: "+" ".join(str(e) for e in ev)+"\nbytes per encoded sample: "+" ".join("%.1f"%p for p in per)+"\nmean of the large launches: %.1f\n"%(sum(p for p,e in zip(per,ev) if e>1e6)/max(1,sum(1 for e in ev if e>1e6)))
open("$OUT/march_$c.txt","w").write(s); print(s)
PY
done
