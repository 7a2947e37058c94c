"""Workload for the PMC passes (tools/measure.sh pmc): builds the bench configuration (bench.py's own arguments), trains
PM_WARM steps, then runs PM_STEPS steps and prints what the counters have to be divided by: the samples the prune march
encoded and the samples rendered in that window, and how many launches of each gather kernel the window holds."""
import gc, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from humanrf_amd import ops
args = bench.parse()
torch.manual_seed(123)
scene, loader, seg, val_cams, capture, frames = bench.build_scene(args, "cuda", 0, 1)
model, eng = bench.build_engine(args, "cuda", 0, 1, loader, frames, seg, 1337)
gc.collect(); gc.freeze()
loader.start_replacer(args.replacements_per_step)
for _ in range(int(os.environ.get("PM_WARM", "600"))):
    eng.train_iteration()
torch.cuda.synchronize()
col = eng.collector
tot0 = col.totals.clone()
ops.TIMER = ops.KernelTimer({"prune_march", "encode4d_fwd_save", "encode4d_bwd_tables"})
n1 = rays = 0
steps = int(os.environ.get("PM_STEPS", "8"))
for _ in range(steps):
    st = eng.train_iteration()
    n1 += st.num_samples; rays += st.num_rays
torch.cuda.synchronize()
t = ops.TIMER.summary(); ops.TIMER = None
d = (col.totals - tot0).cpu().tolist()
loader.stop_replacer()
print("PMC_WINDOW steps %d segments %s march_launches %d encoded %d fwd_launches %d bwd_launches %d rendered %d rays %d"
      % (steps, list(seg), t["prune_march"]["launches"], int(d[1]), t["encode4d_fwd_save"]["launches"],
         t["encode4d_bwd_tables"]["launches"], n1, rays), flush=True)
