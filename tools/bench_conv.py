"""Micro-benchmark of the MFMA conv kernels at the cfg-2 layer shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops

dev = torch.device("cuda:0")
SHAPES = [  # Cin, Cout, K, stride, pad, (D,H,W), transposed
    (64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (128, 64, 64), False),
    (64, 128, (3, 3, 3), (1, 2, 2), (1, 1, 1), (128, 64, 64), False),
    (128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (128, 32, 32), False),
    (256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), (128, 16, 16), False),
    (512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), (128, 16, 16), False),
    (512, 128, (3, 4, 4), (1, 2, 2), (1, 1, 1), (128, 16, 16), True),
    (128, 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), (128, 64, 64), True),
]


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for Cin, Cout, K, s, p, dims, tr in SHAPES:
    x = torch.randn(1, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(Cin, Cout, *K, device=dev) if tr else torch.randn(Cout, Cin, *K, device=dev)) * 0.02
    cfg = ops.ConvCfg(s, p, tr)
    y, _ = ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0)
    out_vox = y.shape[2] * y.shape[3] * y.shape[4]
    T = K[0] * K[1] * K[2]
    flops = 2.0 * (x.shape[2] * x.shape[3] * x.shape[4] if tr else out_vox) * T * Cin * Cout
    dz = torch.randn_like(y)
    tf = timeit(lambda: ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0))
    td = timeit(lambda: ops.conv_dgrad(dz, w, tuple(x.shape[2:]), Cin, 0, cfg))
    tw = timeit(lambda: ops.conv_wgrad(dz, x, None, w, cfg, False))
    print(f"{'T' if tr else 'C'} {Cin:4d}->{Cout:4d} K{K} s{s} {dims}: fwd {tf*1e3:7.2f} ms {flops/tf/1e12:6.1f} TF | "
          f"dgrad {td*1e3:7.2f} ms {flops/td/1e12:6.1f} TF | wgrad {tw*1e3:7.2f} ms {flops/tw/1e12:6.1f} TF", flush=True)
