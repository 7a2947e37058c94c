"""Regime-free comparison of two builds of the training step: ms per step at an IDENTICAL, FROZEN model state.
rays/s = 640 k / (visible samples per ray) / step time, and both the samples per ray and the step time move with the regime a trajectory
has reached, so two bench lines of two builds never compare cleanly. Here one build trains, saves the model (reference state-dict layout:
stable across rounds), and every build under test loads it, sets the learning rate to 0 (Adam runs, nothing moves: the regime stays put)
and times the same seeded steps.
  python tools/stepbench.py train  /tmp/ck.pt [bench.py flags]     train SB_TRAIN steps (default 2000) from the standard initialisation, save
  python tools/stepbench.py measure /tmp/ck.pt [bench.py flags]    load, lr = 0, SB_WARM (30) steps, then SB_ROUNDS (4) x SB_STEPS (100) timed
SB_LIB=tools/_build/libhrf_hip_<tag>.so measures a library variant, SB_SET="a.b=value&c=value" sets engine attributes first.
Run from the root of the tree under test (tools/measure.sh stepbench runs it in this tree and in _r05/, a `git archive` of round 5)."""
import gc
import os
import sys
import time

import torch

ROOT = os.getcwd()
sys.path.insert(0, ROOT)
if os.environ.get("SB_LIB"):      # a tuning variant of the library (make -C humanrf_amd/csrc variant TAG=... EXTRA=-D...)
    import humanrf_amd._lib as _hl
    _hl.LIB_PATH = os.path.join(ROOT, os.environ["SB_LIB"])
import bench  # noqa: E402  (the tree's own bench.py: build_scene / build_engine)

mode, path = sys.argv[1], sys.argv[2]
sys.argv = [sys.argv[0]] + sys.argv[3:]
args = bench.parse()
dev = "cuda"
torch.manual_seed(123)
scene, loader, seg, val_cams, capture, frames = bench.build_scene(args, dev, 0, 1)
model, eng = bench.build_engine(args, dev, 0, 1, loader, frames, seg, 1337)
gc.collect(); gc.freeze()
loader.start_replacer(args.replacements_per_step)
if mode == "train":
    for _ in range(int(os.environ.get("SB_TRAIN", "2000"))):
        eng.train_iteration()
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model.reference_state_dict().items()}, path)
    print("saved", path, flush=True)
else:
    model.load_reference_state_dict(torch.load(path, map_location=dev))
    model._refresh_half()
    eng.lr0 = 0.0                                   # Adam runs, nothing moves
    # SB_SET="collector.spec_margin=1.08;overlap_vector_scatter=False": attributes of the engine (dotted paths) set before timing
    for item in filter(None, os.environ.get("SB_SET", "").replace("&", ";").split(";")):
        name, val = item.split("=", 1)
        obj = eng
        *head, last = name.strip().split(".")
        for h in head:
            obj = getattr(obj, h)
        setattr(obj, last, eval(val))
        print("set", name, "=", getattr(obj, last), flush=True)
    torch.manual_seed(4242)
    for _ in range(int(os.environ.get("SB_WARM", "30"))):
        eng.train_iteration()
    steps = int(os.environ.get("SB_STEPS", "100"))
    # SB_AB="attr.path=A|B": alternate an engine attribute between two values round by round INSIDE this process (rounds of one process
    # agree to ~0.02 ms, two processes of identical settings differ by ~1 %, now and then by 6 %: profiles/r06_stepbench_sweep2.txt)
    ab = os.environ.get("SB_AB", "")
    ab_name, ab_vals = (ab.split("=", 1)[0].strip(), [eval(v) for v in ab.split("=", 1)[1].split("|")]) if ab else (None, [])
    acc = {}

    def ab_pick(rnd):       # A B B A A B B A ...: the rounds of a process are the same seeded steps every time and differ among themselves
        return ((rnd + 1) // 2) % 2 if len(ab_vals) == 2 else rnd % max(len(ab_vals), 1)
    for rnd in range(int(os.environ.get("SB_ROUNDS", "4"))):
        if ab_name:
            obj = eng
            *head, last = ab_name.split(".")
            for h in head:
                obj = getattr(obj, h)
            setattr(obj, last, ab_vals[ab_pick(rnd)])
            for _ in range(5):
                eng.train_iteration()
        torch.cuda.synchronize()
        col = eng.collector
        c0 = (col.march_launches, col.iterations_classic, col.iterations_prefetched) if col is not None else (0, 0, 0)
        t0 = time.perf_counter(); rays = samples = drawn = 0
        for _ in range(steps):
            st = eng.train_iteration()
            rays += st.num_rays; samples += st.num_samples; drawn += st.num_rays_drawn
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = (col.march_launches, col.iterations_classic, col.iterations_prefetched) if col is not None else (0, 0, 0)
        print("round %d: %.3f ms/step  %.2f Mray/s  %.2f samples/ray  %.0f rays/step  %.0f drawn/step  ms per 640k samples %.3f  "
              "march launches/step %.2f  classic iterations %d"
              % (rnd, 1e3 * dt / steps, rays / dt / 1e6, samples / max(rays, 1), rays / steps, drawn / steps,
                 1e3 * dt * 640_000 / max(samples, 1), (c1[0] - c0[0]) / steps, c1[1] - c0[1])
              + ("   [%s = %r]" % (ab_name, ab_vals[ab_pick(rnd)]) if ab_name else ""), flush=True)
        if ab_name:
            acc.setdefault(repr(ab_vals[ab_pick(rnd)]), []).append(1e3 * dt / steps)
    for k, v in acc.items():
        print("AB %s = %s: mean %.3f ms/step over %d rounds (%s)" % (ab_name, k, sum(v) / len(v), len(v), " ".join("%.3f" % x for x in v)), flush=True)
loader.stop_replacer()
