"""Do the two instantiations of k_encode4d_fwd (with and without the per-encoding outputs saved) produce the same composed features?
Round 5: they did not in ~1 of 20 000 elements -- the compiler had folded the last fused multiply-add and the rounding to half into
v_fma_mixlo_f16 in one instantiation only. Prints the number of differing elements for batches of one and of
several temporal segments per wavefront, with and without partially filled tiles; 0 everywhere since enc_pin_f32 (tools/run_r5g.sh)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humanrf_amd import ops
from tests.util import make_model
DEV = "cuda"
m = make_model(DEV, (6, 6), tuple(range(15, 27)), log2_T=15, emb=2, table_scale=0.4)
g = torch.Generator().manual_seed(0)
for n in (700, 704, 64, 4096):
    pos = (torch.rand(n, 3, generator=g) - 0.5).to(DEV)
    for mixed in (True, False):
        if mixed:
            fr = torch.randint(15, 27, (n, 1), generator=g, dtype=torch.int32).to(DEV)
        else:
            fr = (15 + (torch.arange(n) * 12) // n).to(torch.int32).reshape(-1, 1).to(DEV)
        xyzt, seg = m._xyzt_seg(pos, fr)
        m._refresh_half()
        f1, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
        f1 = f1.clone()
        f0, _ = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, False)
        f0 = f0.clone()
        f1b, _ = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
        torch.cuda.synchronize()
        d = (f0 != f1)
        print("n", n, "mixed", mixed, "save-vs-nosave differing", int(d.sum()), "of", d.numel(), "| save twice differing", int((f1 != f1b).sum()),
              "| max abs", float((f0.float() - f1.float()).abs().max()))
        if int(d.sum()):
            idx = d.nonzero()
            print("  first rows/cols:", idx[:12].tolist(), "rows%64:", sorted(set((idx[:, 0] % 64).tolist()))[:20], "levels:", sorted(set((idx[:, 1] // 2).tolist())))
