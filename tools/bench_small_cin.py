"""Thin-input convs: nnU-Net first conv (1 -> 32, IN statistics epilogue) and the FLAVR stem on the teacher's windows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
def t(f, n=5):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, Cin, Cout, K, stride, pad, dims, sm) in [(1, 1, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (128, 128, 128), 2),
                                                 (2, 1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (128, 128, 128), 2),
                                                 (127, 2, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (4, 128, 128), 0),
                                                 (1, 1, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (128, 128, 128), 0)]:
    x = torch.randn(N, Cin, *dims, device=dev)
    x = x.contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, *K, device=dev) * 0.05
    b = torch.zeros(Cout, device=dev)
    cfg = ops.ConvCfg(stride, pad, False)
    f = lambda: ops.conv_forward(x, None, w, b, cfg, ops.ACT_RELU if sm == 0 else 0, 0.0, sm)
    y, st = f()
    ref = torch.nn.functional.conv3d(x[:1].cpu().double(), w.cpu().double(), b.cpu().double(), stride, pad)
    if sm == 0:
        ref = ref.relu()
    err = ((y[:1].cpu().double() - ref).abs().max() / ref.abs().max()).item()
    print(f"N{N} {Cin}->{Cout} {K} {dims}: {t(f):7.3f} ms  err {err:.1e}", flush=True)
