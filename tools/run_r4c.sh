#!/bin/bash
# round-4 check C: GPU test suite; kbench A/B of the unshared fine levels; the rewritten scatter on both models; MFMA chain
# reproducer; fp32 vs fp16 gradient boundaries at the headline regime; the data-parallel step forced onto a one-rank RCCL group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4c}
mkdir -p $OUT
cd $R
export HRF_TEST_DIAG=$OUT/diag.txt
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log | cut -c1-220
echo "== mfma chain reproducer"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_chain_repro.hip -o /tmp/mfma_chain_repro 2>/dev/null && timeout 60 /tmp/mfma_chain_repro | tee $OUT/mfma_chain_repro.txt
export KB_WARM=2000 KB_REPS=20 KB_CACHE=/tmp/kb_r4c.pt
rm -f $KB_CACHE
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for lib in "" tools/_build/libhrf_hip_p13.so tools/_build/libhrf_hip_p11.so tools/_build/libhrf_hip_p8.so; do
  for mode in march fwd; do
    echo "== lib=${lib:-default} mode=$mode"
    KB_LIB=$lib KB_ONLY=$mode timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:"
  done
done
trace() {  # per-kernel times of the binned scatter from a kernel trace
  rm -rf /tmp/kt
  KB_REPS=5 KB_ONLY=scatterprof timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/kbench.py > $OUT/kb_scatter_$1.log 2>&1
  grep -E "ms$|records|batch" $OUT/kb_scatter_$1.log
  f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
  python - <<PY
import csv, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "scatter" in k or "bwd_tables" in k:
        by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-40s n=%d last6 (ms): %s" % (k[:40], len(v), " ".join("%.3f" % x for x in v[-6:])))
PY
}
echo "== scatter, default model"; trace default
export KB_SEGMENTS=50 KB_CACHE=/tmp/kb_r4c_none.pt
rm -f $KB_CACHE
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm_none.log 2>&1
echo "== scatter, one 2^18 segment"; trace none
unset KB_SEGMENTS KB_CACHE
echo "== gradient boundaries: fp32 vs fp16 (camera_embedding_dim 0: no embedding noise in the novel-view PSNR), then the example config"
STEPS=2080 timeout 600 python tools/psnr_variance.py emb0:4 emb0fp16b:4 default:2 fp16b:2 2>&1 | grep -E "run|segments" | tee $OUT/boundaries.txt
echo "== data-parallel step on a one-rank RCCL group"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 \
  --force-collectives --backend nccl --pretrain 500 --steps 40 --no-cpu-baseline --no-validation --curve '' > $OUT/rccl_world1.json 2> $OUT/rccl_world1.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/rccl_world1.json"))
    print("rccl world 1: value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("collectives"), d["config"]["parallelism"][:200])
except Exception as e:
    print("no line:", e); print(open("$OUT/rccl_world1.err").read()[-2500:])
PY
