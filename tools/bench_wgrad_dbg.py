"""Timing of the Winograd weight gradient with parts of the kernel compiled out (REHR_WW_DBG bits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
shapes = [(64, 64, 128, 64, 64), (512, 512, 128, 16, 16)]
for (Cin, Cout, D, H, W) in shapes:
    x = torch.randn(1, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    dz = torch.randn(1, Cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    for rep in range(3):
        for dbg in list(range(16)):
            os.environ["REHR_WW_DBG"] = str(dbg)
            for _ in range(2):
                ops.conv_wgrad(dz, x, None, w, cfg, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv_wgrad(dz, x, None, w, cfg, True)
            e1.record(); torch.cuda.synchronize()
            print(f"{Cin}x{Cout} dbg={dbg:2d}: {e0.elapsed_time(e1)/5*1e3:8.1f} us", flush=True)
