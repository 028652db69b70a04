"""One Winograd-eligible weight gradient, a few launches (for rocprofv3 --pmc passes): Cin Cout D H W"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
Cin, Cout, D, H, W = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
x = torch.randn(1, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
dz = torch.randn(1, Cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
for _ in range(4):
    ops.conv_wgrad(dz, x, None, w, cfg, True)
torch.cuda.synchronize()
