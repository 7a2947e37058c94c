#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE) and L2 atomic requests (TCC_ATOMIC_sum) of the three gather kernels, per unit
# of work, each counter in its own rocprofv3 pass (kernel trace + pmc only). usage: bash tools/run_pmc_r02.sh TAG [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum; do
  rm -rf /tmp/pm_$c
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_prune_march|k_encode4d_fwd|k_encode4d_bwd_tables_lm" --output-format csv -d /tmp/pm_$c -o m -- python $R/tools/pmc_driver.py "$@" > $OUT/run_$c.log 2>&1
  f=$(find /tmp/pm_$c -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, re, collections
log = open("$OUT/run_$c.log").read()
m = re.search(r"PMC_WINDOW steps (\d+) segments (\[.*?\]) march_launches (\d+) encoded (\d+) fwd_launches (\d+) bwd_launches (\d+) rendered (\d+) rays (\d+)", log)
if not m:
    print("no PMC_WINDOW line", log[-400:]); raise SystemExit
steps, segs, ml, enc, fl, bl, n1, rays = m.group(1), m.group(2), *[int(x) for x in m.groups()[2:]]
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r["Counter_Name"] == "$c":
        by[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]))
unit = 1024.0 if "$c".endswith("_SIZE") else 1.0   # FETCH_SIZE / WRITE_SIZE count kilobytes
out = ["# $c, separate pass, window of %s steps, segments %s" % (steps, segs)]
for name, launches, units, what in (("k_prune_march", ml, enc, "encoded sample"), ("k_encode4d_fwd", fl, n1, "rendered sample"),
                                    ("k_encode4d_bwd_tables_lm", bl, n1, "rendered sample")):
    key = [k for k in by if name in k]
    if not key: continue
    vals = by[key[0]][-launches:]
    tot = sum(vals) * unit
    out.append("%-28s launches %3d  total %.6g  units %d  per %s %.2f  per launch %.6g" % (name, launches, tot, units, what, tot / max(units, 1), tot / max(launches, 1)))
open("$OUT/$c.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
done
