"""Micro-benchmark: fused multi-tensor Adam (hrf_adam_multi) against one hrf_adam_step launch per tensor, bench-sized model."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humanrf_amd import ops
from humanrf_amd.trainer import TrainEngine
from tests.util import make_model
dev = "cuda"
m = make_model(dev, (6, 6, 6, 12, 6, 6, 12), tuple(range(15, 65)), log2_T=19, emb=2)
eng = TrainEngine(m, loader=None)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
G = eng.num_groups
def multi(touch):
    def f():
        eng._touched.fill_(0)
        for s in touch: eng._touched[1 + s] = 1
        ops.adam_multi(eng._adam_desc, eng._adam_count, eng.num_groups, eng._adam_total, 1e-2, 0.9, 0.99, 1e-15, 1024.0, eng.opt_state, eng._adam_ws, scaler=eng.scaler)
    return f
def fills(touch):
    def f():
        eng._touched.fill_(0)
        for s in touch: eng._touched[1 + s] = 1
    return f
flags = torch.zeros(1, dtype=torch.int32, device=dev)
def single():
    for p, g, ea, eas in zip(eng._params, eng._grads, eng.exp_avg, eng.exp_avg_sq):
        ops.adam_step(p.data.view(-1), g, ea.view(-1), eas.view(-1), None, 1e-2, 0.9, 0.99, 1e-15, 5, 1024.0, flags)
def single_p16():
    ops.adam_step(m.table_params.data, eng._grads[0], eng.exp_avg[0], eng.exp_avg_sq[0], m._tables_h[:m.table_params.numel()], 1e-2, 0.9, 0.99, 1e-15, 5, 1024.0, flags)
print("params", eng._adam_total, "tensors", eng._adam_count)
print("one launch per tensor (no p16): %.3f ms" % timeit(single))
print("tables only with p16:           %.3f ms" % timeit(single_p16))
for touch in ([0,1,2,3,4,5,6], [0,1,2], [3], []):
    t_f = timeit(fills(touch)); t_m = timeit(multi(touch))
    print("multi, segments %s: %.3f ms (flag fills alone %.3f ms)" % (touch, t_m, t_f))
