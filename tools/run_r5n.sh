#!/bin/bash
# round 5, call N (final sources): PMC traffic passes, a parity subset, the driver's command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5n
mkdir -p $OUT
cd $R
PM_WARM=2000 bash tools/run_pmc_r05.sh r5n_pmc --trials 1 2>&1 | grep -E "per (encoded|rendered) sample|kernel_sources" | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_fixtures.py tests/test_gpu_round5.py -q -m gpu 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_driver_cmd.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "min/max", d["value_min"], d["value_max"])
    for t in d["trials"]: print(t)
    print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_over_algorithmic"])
    print([(k["kernel"][:20], k["frac"], k["traffic_over_algorithmic"], k["ms_per_step"]) for k in d["roofline_kernels"]])
    print(d["validation_psnr_db"], d["cpu_baseline"]["value"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_driver_cmd.err").read()[-2000:])
PY
