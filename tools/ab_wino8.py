"""A/B of the two organisations of the big-tile Winograd kernel (GPU box): 4 waves (one per SIMD) vs 8 waves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops, hip_backend as hb
dev = torch.device("cuda:0")
SHAPES = [(64, 64, (128, 64, 64)), (128, 128, (128, 32, 32)), (256, 256, (128, 16, 16)), (512, 512, (128, 16, 16)),
          (64, 128, (64, 48, 40))]
for Cin, Cout, dims in SHAPES:
    x = torch.randn(1, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    b = torch.randn(Cout, device=dev)
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    res = {}
    for flag in (False, True):
        hb.USE_WINO_8WAVE = flag
        f = lambda: ops.conv_forward(x, None, w, b, cfg, 2, 0.01, 2)
        y, st = f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        res[flag] = (y, st, (time.perf_counter() - t) / 10)
    hb.USE_WINO_8WAVE = False
    dy = (res[True][0] - res[False][0]).abs().max().item() / res[False][0].abs().max().item()
    ds = (res[True][1] - res[False][1]).abs().max().item() / res[False][1].abs().max().item()
    flops = 2.0 * dims[0] * dims[1] * dims[2] * 27 * Cin * Cout
    print(f"{Cin:4d}->{Cout:4d} {dims}: 4-wave {res[False][2]*1e3:7.3f} ms ({flops/res[False][2]/1e12:5.1f} TF)  8-wave {res[True][2]*1e3:7.3f} ms "
          f"({flops/res[True][2]/1e12:5.1f} TF)  max diff y {dy:.1e} stats {ds:.1e}", flush=True)
