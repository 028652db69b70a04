"""Per-block fixed cost of the big-tile Winograd kernel (GPU box, run under rocprofv3 --kernel-trace --stats): the same
lattice and C_out with C_in = 64 / 128 / 256 -> kernel time is a + b * C_in; a / rounds = prologue + epilogue per block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
dev = torch.device("cuda:0")
for Cin in (64, 128, 256):
    for dims, Cout in (((128, 64, 64), 64), ((64, 64, 64), 64)):
        x = torch.randn(1, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
        cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
        for _ in range(5):
            ops.conv_forward(x, None, w, None, cfg, 2, 0.01, 2)
        torch.cuda.synchronize()
