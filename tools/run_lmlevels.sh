#!/bin/bash
# per-level time and L2 atomic requests of the table scatter. usage: bash tools/run_lmlevels.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-lmlevels}
mkdir -p $OUT
KB_WARM=2000 KB_REPS=3 KB_ONLY=lmlevels timeout 300 python $R/tools/kbench.py > $OUT/times.txt 2>&1
cat $OUT/times.txt | tail -20
rm -rf /tmp/pmk
KB_WARM=2000 KB_REPS=1 KB_ONLY=lmlevels timeout 400 rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum --kernel-include-regex "k_encode4d_bwd_tables_lm" --output-format csv -d /tmp/pmk -o m -- python $R/tools/kbench.py > $OUT/pmc.log 2>&1
f=$(find /tmp/pmk -name "*counter_collection.csv" | head -1)
python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$f")) if r["Counter_Name"] == "TCC_ATOMIC_sum"]
vals = [float(r["Counter_Value"]) for r in rows]
# kbench with KB_REPS=1: per level 2 launches (warm + timed), then 2 for all levels; the engine's own 2000 training launches come first
tail = vals[-34:]
out = ["level %2d  atomic requests %.4g" % (l, tail[2 * l + 1]) for l in range(16)] + ["all levels %.4g" % tail[33]]
open("$OUT/atomics.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
