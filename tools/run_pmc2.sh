#!/bin/bash
# FETCH_SIZE calibration on a known random-gather byte count + march traffic per launch.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal -o cal -- $GRAFT_REPO_ROOT/tools/microbench/_build/gather_bench > $OUT/gather_bench.log 2>&1
f=$(find /tmp/cal -name "*counter_collection.csv" | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if r["Counter_Name"]=="FETCH_SIZE"]
out=open("$OUT/calibration.txt","w")
# gather_bench launches each config twice (32 dispatches); lanes per dispatch = 4096*256*64*32
lanes=4096*256*64*32
names=[(fp,g) for fp in ("16KB","1MB","32MB","256MB") for g in (1,2,4,16)]
for i,r in enumerate(rows):
    fp,g=names[i//2]
    kb=float(r["Counter_Value"])
    out.write("footprint %s lanes/line %d : FETCH_SIZE %.0f KB ; lines touched (upper bound) %.3e ; FETCH bytes per lane-load %.2f ; per 64B-line-request %.2f\n"%(fp,g,kb,lanes/g,kb*1024/lanes,kb*1024/(lanes/g)))
out.close()
PY
cat $OUT/calibration.txt | awk 'NR%2==0'
PM_WARM=1000 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_prune_march" --output-format csv -d /tmp/m1 -o m1 -- python $GRAFT_REPO_ROOT/tools/prof_march.py > $OUT/march_run.log 2>&1
grep "march launch" $OUT/march_run.log | tail -8
f=$(find /tmp/m1 -name "*counter_collection.csv" | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if r["Counter_Name"]=="FETCH_SIZE"]
vals=[float(r["Counter_Value"]) for r in rows][-8:]
open("$OUT/march_fetch.txt","w").write("last 8 k_prune_march dispatches FETCH_SIZE (KB): "+" ".join("%.0f"%v for v in vals)+"\n")
print(open("$OUT/march_fetch.txt").read())
PY
PM_WARM=1000 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_prune_march" --output-format csv -d /tmp/m2 -o m2 -- python $GRAFT_REPO_ROOT/tools/prof_march.py > $OUT/march_run2.log 2>&1
f=$(find /tmp/m2 -name "*counter_collection.csv" | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if r["Counter_Name"]=="WRITE_SIZE"]
vals=[float(r["Counter_Value"]) for r in rows][-8:]
open("$OUT/march_write.txt","w").write("last 8 k_prune_march dispatches WRITE_SIZE (KB): "+" ".join("%.0f"%v for v in vals)+"\n")
print(open("$OUT/march_write.txt").read())
PY
grep "march launch" $OUT/march_run2.log | tail -8
