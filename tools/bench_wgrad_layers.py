"""Weight-gradient timing + check against the direct kernels on the Winograd-eligible layer shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import hip_backend, ops
dev = torch.device("cuda:0")
shapes = [(2, 64, 32, 128, 128, 128), (2, 32, 64, 64, 64, 64), (2, 32, 16, 256, 128, 128), (2, 128, 64, 64, 64, 64), (2, 256, 256, 16, 16, 16),
          (1, 64, 64, 128, 64, 64), (1, 128, 128, 128, 32, 32), (1, 512, 512, 128, 16, 16), (2, 32, 32, 128, 128, 128),
          (2, 64, 64, 64, 64, 64), (32, 256, 256, 4, 12, 12), (32, 64, 64, 4, 48, 48), (2, 320, 320, 8, 8, 8)]
for (N, Cin, Cout, D, H, W) in shapes:
    x = torch.randn(N, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    dz = torch.randn(N, Cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    hip_backend.USE_WINOGRAD_WGRAD = False
    ref, rb = ops.conv_wgrad(dz, x, None, w, cfg, True)
    hip_backend.USE_WINOGRAD_WGRAD = True
    got, gb = ops.conv_wgrad(dz, x, None, w, cfg, True)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    errb = ((gb - rb).abs().max() / rb.abs().max()).item()
    for _ in range(2):
        ops.conv_wgrad(dz, x, None, w, cfg, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_wgrad(dz, x, None, w, cfg, True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    fl = 2.0 * N * D * H * W * 27 * Cin * Cout
    print(f"N{N} {Cin}->{Cout} {D}x{H}x{W}: {t*1e3:8.1f} us {fl/t/1e9:6.1f} TF  err {err:.2e} bias {errb:.2e}", flush=True)
