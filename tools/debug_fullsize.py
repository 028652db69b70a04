"""How ill-conditioned are the full-size cfg-2 gradients?  (a) Winograd vs direct kernels, (b) direct vs direct with
the input perturbed at fp32 rounding level (1e-7 relative), (c) only the forward/dgrad or only the wgrad Winograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from rehrseg_amd import hip_backend
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from test_full_size_gpu import direct_kernels, _grads
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNet_3D_3D(1, "unet_18", 128, 4).to(dev)
g = torch.Generator().manual_seed(0)
x = torch.rand(1, 1, 128, 128, 128, generator=g).to(dev)
tgt = torch.rand(1, 1, 4, 128, 128, generator=g).to(dev)
xin = [x]
loss_fn = lambda: ((model(xin[0].clone()) - tgt) ** 2).mean()
_, g_w = _grads(model, loss_fn)
with direct_kernels():
    _, g_d = _grads(model, loss_fn)
    xin[0] = x * (1 + 1e-7 * torch.randn_like(x))
    _, g_p = _grads(model, loss_fn)
    xin[0] = x
os.environ["REHR_WINO_WGRAD"] = "0"; os.environ["REHR_WINO22"] = "0"      # Winograd fwd/dgrad (3x3 only), direct wgrad
_, g_f = _grads(model, loss_fn)
os.environ.pop("REHR_WINO_WGRAD"); os.environ.pop("REHR_WINO22")
hip_backend.USE_WINOGRAD = False                                            # direct fwd/dgrad, Winograd wgrad
_, g_g = _grads(model, loss_fn)
hip_backend.USE_WINOGRAD = True
rel = lambda a, b: float((a - b).norm() / b.norm())
for n in g_d:
    if "attn" in n or n.endswith("bias"): continue
    print(f"{n:40s} wino-vs-direct {rel(g_w[n], g_d[n]):.1e} | input*(1+1e-7) {rel(g_p[n], g_d[n]):.1e} | fwd/dgrad-only {rel(g_f[n], g_d[n]):.1e} | wgrad-only {rel(g_g[n], g_d[n]):.1e}")
