"""One layer on the pipelined 32-channel-tile kernel, for counter runs (GPU box): python tools/run_w32p_once.py [0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops, hip_backend as hb
hb.USE_W32_PIPELINED = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
dev = torch.device("cuda:0")
x = torch.randn(2, 32, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.02
cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
for _ in range(4):
    ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 2)
torch.cuda.synchronize()
