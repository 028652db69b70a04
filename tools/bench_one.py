"""Run one conv forward repeatedly (for rocprofv3 --pmc runs on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops
Cin, Cout, D, H, W, reps = (int(a) for a in sys.argv[1:7])
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
dev = torch.device("cuda:0")
x = torch.randn(1, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
y, _ = ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0)
dz = torch.randn_like(y)
for _ in range(reps):
    if mode == "fwd":
        ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0)
    elif mode == "wgrad":
        ops.conv_wgrad(dz, x, None, w, cfg, False)
torch.cuda.synchronize()
