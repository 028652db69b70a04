import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
for Cin, Cout, dims in [(512, 512, (128, 32, 16)), (64, 64, (128, 64, 64)), (128, 128, (128, 32, 32)), (512, 512, (128, 32, 16))]:
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
    fl = 2.0 * dims[0] * dims[1] * dims[2] * 27 * Cin * Cout
    print(f"{Cin}->{Cout} {dims}: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF", flush=True)
