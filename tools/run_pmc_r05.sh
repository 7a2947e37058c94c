#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE) and memory-side atomic requests (TCC_ATOMIC_sum) of the gather / scatter
# kernels per unit of work, each counter in its own rocprofv3 pass (kernel trace + pmc only), plus MfmaUtil-style
# counters of the MLP kernels. Writes gpurun_out/$TAG/{FETCH_SIZE,WRITE_SIZE,TCC_ATOMIC_sum}.txt and traffic.json (with
# the fingerprint of the kernel sources, bench.kernel_source_fingerprint). usage: bash tools/run_pmc_r05.sh TAG [bench args]
# Round 5: FETCH_SIZE is corrected per kernel by the factors calibrated on known byte counts (profiles/r05_pmc_fetch_write_calibration.txt:
# the counter reports half of a coalesced streaming read of any width, all of a random gather).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
KERN="k_prune_march|k_encode4d_fwd|k_scatter_emit|k_scatter_accumulate|k_encode4d_bwd_tables_lm|k_encode4d_bwd_vectors"
for c in FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum; do
  rm -rf /tmp/pm_$c
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$KERN" --output-format csv -d /tmp/pm_$c -o m -- python $R/tools/pmc_driver.py "$@" > $OUT/run_$c.log 2>&1
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
out = ["# $c, separate pass, window of %s steps, segments %s, %d rays, %d rendered samples" % (steps, segs, rays, n1)]
for name, launches, units, what in (("k_prune_march", ml, enc, "encoded sample"), ("k_encode4d_fwd", fl, n1, "rendered sample"),
                                    ("k_scatter_emit", bl, n1, "rendered sample"), ("k_scatter_accumulate", bl, n1, "rendered sample"),
                                    ("k_encode4d_bwd_tables_lm", bl, n1, "rendered sample"), ("k_encode4d_bwd_vectors", bl, n1, "rendered sample")):
    key = [k for k in by if name in k]
    if not key: continue
    vals = by[key[0]][-launches:]
    tot = sum(vals) * unit
    out.append("%-28s launches %3d  total %.6g  units %d  per %s %.2f  per launch %.6g" % (name, launches, tot, units, what, tot / max(units, 1), tot / max(launches, 1)))
open("$OUT/$c.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
done
python - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
def per(counter, kernel):
    try:
        for line in open("$OUT/%s.txt" % counter):
            if line.startswith(kernel + " "):
                return float(re.search(r"per (?:encoded|rendered) sample ([0-9.eE+-]+)", line).group(1))
    except FileNotFoundError:
        pass
    return None
FETCH_CORRECTION = {"k_scatter_accumulate": 2.0, "k_scatter_emit": 2.0, "k_encode4d_bwd_vectors": 2.0}   # coalesced streams: x 2
def both(k):
    f, w = per("FETCH_SIZE", k), per("WRITE_SIZE", k)
    if f is None or w is None:
        return None
    c = FETCH_CORRECTION.get(k, 1.0)
    return {"fetch_bytes_per_encoded_sample": f * c, "write_bytes_per_encoded_sample": w, "fetch_size_as_reported": f, "fetch_correction": c}
j = {"kernel_sources_sha256": bench.kernel_source_fingerprint(),
     "source": "tools/run_pmc_r05.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC_sum, one counter per pass, "
               "8-step window of the default bench configuration after PM_WARM training steps; FETCH_SIZE corrected per kernel "
               "(x 2 for the kernels whose reads are coalesced streams, x 1 for the gather kernels: "
               "profiles/r05_pmc_fetch_write_calibration.txt)",
     "header": open("$OUT/FETCH_SIZE.txt").readline().strip()}
for k in ("k_prune_march", "k_encode4d_fwd"):
    b = both(k)
    if b: j[k] = b
e, a = both("k_scatter_emit"), both("k_scatter_accumulate")
if e and a:
    j["table_scatter"] = {"fetch_bytes_per_encoded_sample": e["fetch_bytes_per_encoded_sample"] + a["fetch_bytes_per_encoded_sample"],
                          "write_bytes_per_encoded_sample": e["write_bytes_per_encoded_sample"] + a["write_bytes_per_encoded_sample"],
                          "l2_atomic_requests_per_sample": (per("TCC_ATOMIC_sum", "k_scatter_emit") or 0.0) + (per("TCC_ATOMIC_sum", "k_scatter_accumulate") or 0.0),
                          "kernels": {"k_scatter_emit": e, "k_scatter_accumulate": a}}
v = both("k_encode4d_bwd_vectors")
if v:
    v["l2_atomic_requests_per_sample"] = per("TCC_ATOMIC_sum", "k_encode4d_bwd_vectors")
    j["k_encode4d_bwd_vectors"] = v
json.dump(j, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(j, indent=1))
PY
