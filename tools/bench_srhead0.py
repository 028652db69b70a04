"""sr_head.0 (32 -> 16, 3x3x3, models/seg_model.py:197) forward / dgrad / wgrad timing at the cfg-3 shape (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
N, Cin, Cout, D, H, W = 2, 32, 16, 512, 128, 128
x = torch.randn(N, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(Cout, device=dev)
cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
y, _ = ops.conv_forward(x, None, w, b, cfg, ops.ACT_RELU, 0.0, 0)
dy = torch.randn_like(y)
fl = 2.0 * N * D * H * W * 27 * Cin * Cout
def t(f, n=5):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, f in [("fwd", lambda: ops.conv_forward(x, None, w, b, cfg, ops.ACT_RELU, 0.0, 0)),
                ("dgrad", lambda: ops.conv_dgrad(dy, w, (D, H, W), Cin, 0, cfg)),
                ("wgrad", lambda: ops.conv_wgrad(dy, x, None, w, cfg, True))]:
    ms = t(f)
    print(f"{name:6s} {ms:7.3f} ms  {fl/ms/1e9:6.1f} TF (algorithmic)", flush=True)
xs = x[:1, :, :8, :32, :32].contiguous(memory_format=torch.channels_last_3d)
ys, _ = ops.conv_forward(xs, None, w, b, cfg, 0, 0.0, 0)
ref = torch.nn.functional.conv3d(xs.cpu().double(), w.cpu().double(), b.cpu().double(), 1, 1)
print("fwd err", ((ys.cpu().double() - ref).abs().max() / ref.abs().max()).item())
dys = torch.randn_like(ys)
xr = xs.cpu().double().requires_grad_(True); wr = w.cpu().double().requires_grad_(True)
torch.nn.functional.conv3d(xr, wr, None, 1, 1).backward(dys.cpu().double())
dws, dbs = ops.conv_wgrad(dys, xs, None, w, cfg, True)
dxs, _ = ops.conv_dgrad(dys, w, (8, 32, 32), Cin, 0, cfg)
print("wgrad err", ((dws.cpu().double() - wr.grad).abs().max() / wr.grad.abs().max()).item(),
      "dgrad err", ((dxs.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max()).item())
