import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import segmodel_oracle as so
from oracle.detinit import det_input
from test_segmodel_cpu import build
cfg = so.ANISO_PLAN
shape = (1, 1, 16, 96, 96)
m, sd = build(cfg, "cuda:0")
x = det_input("seg.x", shape, "randn")
out, out_up, skips = m(x.clone().cuda(), return_inetermediate_feature=True)
g1, g2, g3 = (det_input(n, tuple(t.shape), "randn") for n, t in (("g1", out), ("g2", out_up), ("g3", skips[1])))
((out * g1.cuda()).mean() + (out_up * g2.cuda()).mean() + (skips[1] * g3.cuda()).mean()).backward()
res = {}
for dt in (torch.float64, torch.float32):
    osd = {k: v.to(dt).requires_grad_() for k, v in sd.items() if k in so.segmodel_shapes(cfg)}
    r_out, r_up, r_skips = so.seg_model(osd, x.to(dt), cfg, return_features=True)
    ((r_out * g1.to(dt)).mean() + (r_up * g2.to(dt)).mean() + (r_skips[1] * g3.to(dt)).mean()).backward()
    res[dt] = (osd, r_out, r_up, r_skips)
o64, o32 = res[torch.float64], res[torch.float32]
def rel(a, b): return float((a.detach().cpu().double() - b.detach().double()).norm() / (b.detach().double().norm() + 1e-30))
print("out", rel(out, o64[1]), "oracle32:", rel(o32[1], o64[1]))
print("up ", rel(out_up, o64[2]), "oracle32:", rel(o32[2], o64[2]))
for i, (a, b, c) in enumerate(zip(skips, o64[3], o32[3])):
    print("skip", i, rel(a, b), "oracle32:", rel(c, b))
params = dict(m.named_parameters())
rows = []
for k, v in o64[0].items():
    if v.grad is None or 'conv.bias' in k: continue
    rows.append((rel(params[k].grad, v.grad), rel(o32[0][k].grad, v.grad), k, float(v.grad.norm())))
for r in sorted(rows, reverse=True)[:25]:
    print(f"{r[0]:.3e}  oracle32 {r[1]:.3e}  |g|={r[3]:.3e}  {r[2]}")
