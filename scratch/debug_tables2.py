import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hrf_oracle as O
from tests.util import make_model, oracle_model_from
from humanrf_amd import ops
from humanrf_amd.dataset.input_batch import InputBatch
from humanrf_amd.volume_rendering import render
DEV = "cuda"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "hotpath_seed123.npz")))
m = make_model(DEV, (6,), tuple(range(15, 21)), log2_T=12, emb=2, seed=1337, table_scale=0.3)
om = oracle_model_from(m, requires_grad=True)
o = torch.from_numpy(g["smp_origins"]); d = torch.from_numpy(g["smp_dirs"]); fr = torch.from_numpy(g["smp_frames_s"]); cm = torch.from_numpy(g["smp_cams_s"])
t = torch.from_numpy(g["smp_t"]).clone(); ray = torch.from_numpy(g["smp_ray"]).long(); jit = torch.from_numpy(g["jitter"])
vis = torch.from_numpy(g["prune_vis"]); bg = torch.from_numpy(g["background"])
tj = (t + jit * 4e-4)[vis].view(-1, 1); r1 = ray[vis]
color, acc = O.render(om, o, d, fr, cm, tj, r1, bg, True)
dC, dA = torch.from_numpy(g["d_color"]), torch.from_numpy(g["d_acc"])
torch.autograd.backward([color, acc], [dC, dA])
ib = InputBatch(ray_origins=o.to(DEV), ray_directions=d.to(DEV), rgba=None, frame_numbers=fr.to(DEV).view(-1, 1), camera_numbers=cm.to(DEV).view(-1, 1),
                sample_distances=tj.to(DEV), ray_indices=r1.to(DEV), unique_frame_numbers=fr[:1].to(DEV).view(-1, 1))
ro = render(ib, m, bg.to(DEV), True)
print("color err", float((ro.color.detach().cpu() - color.detach()).abs().max()))
torch.autograd.backward([ro.color, ro.weights_sum], [dC.to(DEV) * 65536.0, dA.to(DEV) * 65536.0])
ent = m.entries_per_segment[0]
tg = (m.table_params.grad.view(4, ent, 2).cpu() / 65536.0)
for e in range(4):
    ref = om.tables[0][e].grad
    a = tg[e].double().reshape(-1); b = ref.double().reshape(-1)
    print("enc", e, "cos", float(a @ b / (a.norm() * b.norm())), "rel", float((a - b).norm() / b.norm()), "norms", float(a.norm()), float(b.norm()))
    for l in range(16):
        lv = om.levels[0][l]
        a = tg[e][lv.offset:lv.offset + lv.size].double().reshape(-1); b = ref[lv.offset:lv.offset + lv.size].double().reshape(-1)
        print("   level", l, "cos %.6f" % float(a @ b / (a.norm() * b.norm() + 1e-300)), "norms %.4e %.4e" % (float(a.norm()), float(b.norm())))
a = (m.vectors.grad[0].double().cpu() / 65536.0).reshape(-1); b = om.vectors[0].grad.double().reshape(-1)
print("vectors cos", float(a @ b / (a.norm() * b.norm())), float(a.norm()), float(b.norm()))
sw = torch.cat([w.grad.reshape(-1) for w in om.sigma_w]).double(); a = m.sigma_params.grad.double().cpu() / 65536
print("sigma cos", float(a @ sw / (a.norm() * sw.norm())), float(a.norm()), float(sw.norm()))
