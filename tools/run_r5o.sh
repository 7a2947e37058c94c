#!/bin/bash
# round 5, call O: the paper's setting (--emb 0: the PSNR half of the metric without embedding noise) and the one-segment model (2^18-entry tables).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5o
mkdir -p $OUT
cd $R
timeout 600 python bench.py --emb 0 --trials 1 --no-cpu-baseline > $OUT/config_emb0.json 2> $OUT/config_emb0.err
timeout 600 python bench.py --partitioning none --trials 1 --no-cpu-baseline --curve '' > $OUT/config_partitioning_none.json 2> $OUT/config_partitioning_none.err
python - <<PY
import json
for n in ("config_emb0", "config_partitioning_none"):
    try:
        d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print(n, "value %.0f ms/step %.3f spr %.2f val psnr %s train psnr %s" % (d["value"], d["ms_per_step"], d["samples_per_ray_post"], d.get("validation_psnr_db"), d["train_psnr_db"]),
              [(p["steps_trained_before"], p["rays_per_s_this_rank"], p.get("validation_psnr_db")) for p in d["regime_curve"]],
              [(k["kernel"][:16], k["frac"], k["ms_per_step"]) for k in d["roofline_kernels"]])
    except Exception as e:
        print(n, "no line:", e); print(open("$OUT/%s.err" % n).read()[-1500:])
PY
