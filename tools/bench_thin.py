"""Thin-output convolution (sr_head.2: 16 -> 2, 5x5x5, models/seg_model.py:199) forward / dgrad / wgrad timing (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
N, Cin, Cout, K, D, H, W = 2, 16, 2, 5, 512, 128, 128
if len(sys.argv) > 1:
    N = int(sys.argv[1])
x = torch.randn(N, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(Cout, Cin, K, K, K, device=dev) * 0.02
b = torch.zeros(Cout, device=dev)
cfg = ops.ConvCfg((1, 1, 1), (K // 2,) * 3, False)
y, _ = ops.conv_forward(x, None, w, b, cfg, 0, 0.0, 0)
dy = torch.randn_like(y)
fl = 2.0 * N * D * H * W * K ** 3 * Cin * Cout
def t(f, n=5):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, f in [("fwd", lambda: ops.conv_forward(x, None, w, b, cfg, 0, 0.0, 0)),
                ("dgrad", lambda: ops.conv_dgrad(dy, w, (D, H, W), Cin, 0, cfg)),
                ("wgrad", lambda: ops.conv_wgrad(dy, x, None, w, cfg, True))]:
    ms = t(f)
    print(f"{name:6s} {ms:7.3f} ms  {fl/ms/1e9:6.1f} TF", flush=True)
# spot check against torch on a small crop
xs = x[:1, :, :12, :20, :24].contiguous(memory_format=torch.channels_last_3d)
ys, _ = ops.conv_forward(xs, None, w, b, cfg, 0, 0.0, 0)
ref = torch.nn.functional.conv3d(xs.cpu().double(), w.cpu().double(), b.cpu().double(), 1, K // 2)
print("fwd err", ((ys.cpu().double() - ref).abs().max() / ref.abs().max()).item())
dys = torch.randn_like(ys)
xr = xs.cpu().double().requires_grad_(True); wr = w.cpu().double().requires_grad_(True); br = b.cpu().double().requires_grad_(True)
torch.nn.functional.conv3d(xr, wr, br, 1, K // 2).backward(dys.cpu().double())
dws, dbs = ops.conv_wgrad(dys, xs, None, w, cfg, True)
dxs, _ = ops.conv_dgrad(dys, w, (12, 20, 24), Cin, 0, cfg)
print("wgrad err", ((dws.cpu().double() - wr.grad).abs().max() / wr.grad.abs().max()).item(),
      "bias", ((dbs.cpu().double() - br.grad).abs().max() / br.grad.abs().max()).item(),
      "dgrad err", ((dxs.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max()).item())
