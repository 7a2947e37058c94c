#!/bin/bash
# round-4 final artefacts: GPU test suite; default bench line + rocprofv3 kernel stats + timed region; PMC traffic with the kernel
# source fingerprint; the one-segment (2^18) and 100-frame-segment (2^19) configurations with the march's FETCH_SIZE; 10 000-step
# curves of the 250- and 1 000-frame shapes. usage: bash tools/run_r4e.sh TAG [stages...] (default: all)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4e}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
STAGES=${@:-"tests profile pmc configs fetch curves"}
for st in $STAGES; do
case $st in
tests)
  timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
  tail -6 $OUT/pytest.log | cut -c1-200 ;;
profile)
  timeout 600 bash tools/run_profile.sh $TAG/profile > $OUT/profile.log 2>&1
  python - <<PY
import json
d = json.load(open("$OUT/profile/bench_plain.json"))
print("default: value %.0f ms/step %.3f spr %.2f train psnr %.2f val %s" % (d["value"], d["ms_per_step"], d["samples_per_ray_post"], d["train_psnr_db"], d.get("validation", {}).get("psnr_db_mean")))
for k in d["roofline_kernels"]: print("  ", k["kernel"][:44], k["bound"], k["frac"], k["ms_per_step"])
print("  curve", [(p["steps_trained_before"], p["rays_per_s_this_rank"], p.get("validation_psnr_db")) for p in d["regime_curve"]])
PY
  head -45 $OUT/profile/timed_region.txt ;;
pmc)
  PM_WARM=2000 timeout 600 bash tools/run_pmc_r03.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -40 $OUT/pmc.log | cut -c1-200 ;;
configs)
  for cfg in "none:--partitioning none" "seg100:--partitioning fixed --segment-size 100 --frames 100" "image3008:--image 3008 --pretrain 1000"; do
    name=${cfg%%:*}; a=${cfg#*:}
    timeout 300 python bench.py --no-cpu-baseline --curve '' --steps 60 $a > $OUT/config_$name.json 2> $OUT/config_$name.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/config_$name.json"))
    print("$name: value %.0f ms/step %.3f spr %.2f psnr %.2f val %s" % (d["value"], d["ms_per_step"], d["samples_per_ray_post"], d["train_psnr_db"], d.get("validation", {}).get("psnr_db_mean")), d["kernel_ms_per_step"])
except Exception as e:
    print("$name: no line", e); print(open("$OUT/config_$name.err").read()[-1200:])
PY
  done ;;
fetch)
  for c in FETCH_SIZE; do
    rm -rf /tmp/pmn
    PM_WARM=1500 timeout 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_prune_march|k_encode4d_fwd|k_scatter_emit|k_scatter_accumulate" --output-format csv -d /tmp/pmn -o m -- python tools/pmc_driver.py --partitioning none > $OUT/pmc_none_$c.log 2>&1
    f=$(find /tmp/pmn -name "*counter_collection.csv" | head -1)
    python - <<PY | tee $OUT/pmc_none_$c.txt
import csv, re, collections
log = open("$OUT/pmc_none_$c.log").read()
m = re.search(r"PMC_WINDOW steps (\d+) segments (\[.*?\]) march_launches (\d+) encoded (\d+) fwd_launches (\d+) bwd_launches (\d+) rendered (\d+) rays (\d+)", log)
if not m: print("no PMC_WINDOW line", log[-600:]); raise SystemExit
ml, enc, fl, bl, n1 = int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7))
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r["Counter_Name"] == "$c": by[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]))
print("# $c, --partitioning none (one 2^18 segment), window of %s steps, segments %s" % (m.group(1), m.group(2)))
for name, launches, units in (("k_prune_march", ml, enc), ("k_encode4d_fwd", fl, n1), ("k_scatter_emit", bl, n1), ("k_scatter_accumulate", bl, n1)):
    key = [k for k in by if name in k]
    if key:
        tot = sum(by[key[0]][-launches:]) * 1024.0
        print("%-24s launches %3d  bytes per sample %.1f" % (name, launches, tot / max(units, 1)))
PY
  done ;;
curves)
  for cfg in "frames250:--frames 250" "frames1000:--frames 1000"; do
    name=${cfg%%:*}; a=${cfg#*:}
    STEPS=10000 EVERY=2500 EXTRA="$a" timeout 400 python tools/long_run.py 2>&1 | grep -E "step|segments" | cut -c1-260 | tee $OUT/curve_$name.txt
  done ;;
esac
done
