import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hrf_oracle as O
from tests.util import make_model, oracle_model_from
from humanrf_amd import ops
DEV = "cuda"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "hotpath_seed123.npz")))
m = make_model(DEV, (6,), tuple(range(15, 21)), log2_T=12, emb=2, seed=1337, table_scale=0.3)
om = oracle_model_from(m, requires_grad=False)
o = torch.from_numpy(g["smp_origins"]); d = torch.from_numpy(g["smp_dirs"]); fr = torch.from_numpy(g["smp_frames_s"]); cm = torch.from_numpy(g["smp_cams_s"])
t = torch.from_numpy(g["smp_t"]).clone(); ray = torch.from_numpy(g["smp_ray"]).long(); jit = torch.from_numpy(g["jitter"])
vis = torch.from_numpy(g["prune_vis"]); bg = torch.from_numpy(g["background"])
tj = (t + jit * 4e-4)[vis].view(-1, 1); r1 = ray[vis]
pos = o[r1] + tj * d[r1]
R = o.shape[0]
feats = O.model_features(om, pos, fr[r1]).detach().requires_grad_()
h = O.mlp(feats, om.sigma_w, "None"); h.retain_grad()
sigma = O.truncated_exp(h[:, 0]) * om.density_scale; sigma.retain_grad()
emb = om.camera_embeddings[cm[r1].long()]
x = O.color_net_input(d[r1], h[:, 1:], emb)
rgb = O.mlp(x, om.color_w, "Sigmoid")[:, :3]; rgb.retain_grad()
w = O.render_weight_from_density(tj, tj + 4e-4, sigma, r1)
color = O.accumulate_along_rays(w, r1, rgb, R); acc = O.accumulate_along_rays(w, r1, None, R)
color = color + bg * (1 - acc)
dC, dA = torch.from_numpy(g["d_color"]), torch.from_numpy(g["d_acc"])
torch.autograd.backward([color, acc], [dC, dA])
# device
S = 128.0 * 65536.0
xyzt, seg = m._xyzt_seg(pos.to(DEV), fr[r1].to(DEV).view(-1, 1))
f_d, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, 1, True)
sw1, sw2 = m._sigma_w(); cw1, cw2, cw3 = m._color_w()
for S in (128.0 * 65536.0, 65536.0, 1024.0, 16.0):
    gs = torch.zeros(m.sigma_params.numel(), device=DEV); gc = torch.zeros(m.color_params.numel(), device=DEV); ge = torch.zeros_like(m.camera_embeddings.weight)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    kin = m.color_in_pad
    d_f = ops.mlp_bwd(f_d, d.to(DEV), r1.to(DEV), m.camera_embeddings.weight.detach(), cm.to(DEV), 2, True, sw1, sw2, cw1, cw2, cw3, 100.0,
                      (rgb.grad * S).to(DEV).contiguous(), (sigma.grad * S).to(DEV).contiguous(), gs[:2048], gs[2048:], gc[:64 * kin], gc[64 * kin:64 * kin + 4096], gc[64 * kin + 4096:], ge, flags)
    a = (d_f.cpu().double() / S); b = feats.grad.double()
    print("S=%g flags=%d  dY cos %.6f rel %.4f | max|b| %.3e  scaled: max %.3e median %.3e" % (S, int(flags), float((a * b).sum() / (a.norm() * b.norm())), float((a - b).norm() / b.norm()), float(b.abs().max()), float((b * S).abs().max()), float((b * S).abs().median())))
    per = ((a - b).norm(dim=1) / (b.norm(dim=1) + 1e-30))
    print("   per-sample rel err: median %.4f  p90 %.4f  max %.4f" % (float(per.median()), float(per.quantile(0.9)), float(per.max())))
print("d_sigma*sigma scaled stats:", float((sigma.grad * sigma.detach()).abs().max()), float((sigma.grad * sigma.detach()).abs().median()), " d_rgb max", float(rgb.grad.abs().max()))
print("h.grad[:,0] max/median", float(h.grad[:, 0].abs().max()), float(h.grad[:, 0].abs().median()), "geo grad max/median", float(h.grad[:, 1:].abs().max()), float(h.grad[:, 1:].abs().median()))
