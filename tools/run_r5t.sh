#!/bin/bash
# round 5, call T: the whole GPU test suite, smoke() and the driver's bench command on the final tree.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5t
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
T0=$SECONDS; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench wall $((SECONDS - T0)) s" | tee $OUT/bench_time.txt
python - <<PY
import json
d = json.loads(open("$OUT/bench_driver_cmd.json").read().strip().splitlines()[-1])
print(d["value"], d["value_min"], d["value_max"], d["ms_per_step"], d["samples_per_ray_post"], d["roofline"]["frac"], [k["frac"] for k in d["roofline_kernels"]])
PY
