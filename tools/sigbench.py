"""Accumulate half of the table-gradient scatter on one synthetic batch of 640 k samples (the bench model's seven segments):
one launch / one launch per group / one signalled launch (hrf_scatter_accumulate_signalled), each timed with events over REPS
repetitions after the same emit. KB_LIB=tools/_build/libhrf_hip_<tag>.so measures a library variant (make variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanrf_amd._lib as _hl
if os.environ.get("KB_LIB"):
    _hl.LIB_PATH = os.path.join(ROOT, os.environ["KB_LIB"])
import torch
from humanrf_amd import ops
from tests.test_gpu_scatter import _bench_model, _ray_samples, DEV

REPS = int(os.environ.get("REPS", "20"))
model = _bench_model()
xyzt, seg = _ray_samples(model, 40_000, 16, seed=3)
n = xyzt.shape[0]
g = torch.Generator(device=DEV).manual_seed(4)
dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
vectors = model.vectors.detach()
S = model.num_segments
ws = ops.ScatterWorkspace(n + 1024, S, model.max_level_entries, DEV)
flags = torch.zeros(1, dtype=torch.int32, device=DEV)
out = torch.zeros(model.table_params.numel(), device=DEV)
ops.scatter_emit(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, out, ws, grad_boundary=128.0)
done = torch.zeros(8, dtype=torch.int64, device=DEV)
groups4 = [[0, 1, 2], [3], [4, 5], [6]]


def timeit(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-52s %.3f ms" % (label, a.elapsed_time(b) / REPS), flush=True)


def per_group():
    for grp in groups4:
        ops.scatter_accumulate(model._seg_meta, S, out, ws, flags=flags, seg_first=grp[0], seg_count=len(grp))


print(os.environ.get("KB_LIB", "default library"), "n =", n)
timeit(lambda: ops.scatter_accumulate(model._seg_meta, S, out, ws, flags=flags), "one launch (segments that own tiles)")
timeit(lambda: ops.scatter_accumulate(model._seg_meta, S, out, ws, flags=flags, seg_first=0, seg_count=S), "one launch (every segment by id)")
timeit(per_group, "four launches (one per group)")
if ops.can_stream_wait_value():
    timeit(lambda: ops.scatter_accumulate_signalled(model._seg_meta, S, out, ws, flags, groups4, done), "one launch, signalled per group (4 groups)")
    timeit(lambda: ops.scatter_accumulate_signalled(model._seg_meta, S, out, ws, flags, [[s] for s in range(S)], done), "one launch, signalled per group (7 groups)")

# a checksum of the table gradients of one emit + accumulate from zero: equal between library builds = the same sums to the bit
import hashlib
chk = torch.zeros(model.table_params.numel(), device=DEV)
ops.scatter_emit(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, chk, ws, grad_boundary=128.0)
ops.scatter_accumulate(model._seg_meta, S, chk, ws, flags=flags)
torch.cuda.synchronize()
print("table gradients sha256", hashlib.sha256(chk.cpu().numpy().tobytes()).hexdigest()[:16], "abs sum %.9e" % float(chk.double().abs().sum()),
      "flags", int(flags))
