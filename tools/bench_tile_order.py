"""fp32 Winograd forward convs of the cfg-3 / cfg-2 layer shapes: band-major tile order against slice-major (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import hip_backend as hb, ops
dev = torch.device("cuda:0")

def t(f, n=10):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (N, Cin, Cout, dims, K) in [(2, 32, 32, (128, 128, 128), (3, 3, 3)), (2, 64, 32, (128, 128, 128), (3, 3, 3)),
                                (2, 64, 64, (64, 64, 64), (3, 3, 3)), (2, 128, 128, (32, 32, 32), (3, 3, 3)),
                                (1, 64, 64, (128, 64, 64), (3, 3, 3)), (1, 128, 128, (128, 32, 32), (3, 3, 3)),
                                (1, 512, 512, (128, 16, 16), (3, 3, 3)), (1, 32, 32, (160, 160, 160), (1, 3, 3))]:
    x = torch.randn(N, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, *K, device=dev) * 0.02
    b = torch.zeros(Cout, device=dev)
    cfg = ops.ConvCfg((1, 1, 1), tuple(k // 2 for k in K), False)
    line = f"{N}x{Cin}->{Cout} {dims} k{K}"
    for flag in (True, False):
        hb.WINO_BAND_MAJOR = flag
        ms = t(lambda: ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 2))
        line += f"  {'band' if flag else 'slice'}-major {ms * 1e3:8.1f} us"
    print(line, flush=True)
