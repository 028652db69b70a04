"""Forward timing of the Winograd-eligible cfg-2 layer shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(64, 64, (128, 64, 64)), (128, 128, (128, 32, 32)), (256, 256, (128, 16, 16)), (512, 512, (128, 16, 16)),
          (32, 32, (128, 128, 128))]
for Cin, Cout, dims in SHAPES:
    x = torch.randn(1, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    f = lambda: ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0)
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 10
    flops = 2.0 * dims[0] * dims[1] * dims[2] * 27 * Cin * Cout
    print(f"{Cin:4d}->{Cout:4d} {dims}: fwd {t*1e3:7.3f} ms {flops/t/1e12:6.1f} TF", flush=True)
