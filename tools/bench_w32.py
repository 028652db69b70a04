"""32-channel-tile Winograd layers of cfg-3 (stage 0 / last decoder stage at 128^3), 16-wave against pipelined (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops, hip_backend as hb
dev = torch.device("cuda:0")
def t(f, n=10):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for Cin, Cout, dims in [(32, 32, (2, 128, 128, 128)), (64, 32, (2, 128, 128, 128)), (32, 16, (2, 128, 128, 128))]:
    N, D, H, W = dims
    x = torch.randn(N, Cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    fl = 2.0 * N * D * H * W * 27 * Cin * Cout
    for flag, blocks in ((False, 0), (True, 2), (True, 1), (False, 0), (True, 2), (True, 1)):
        hb.USE_W32_PIPELINED, hb.W32P_BLOCKS = flag, blocks
        tf = t(lambda: ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 2))
        what = "16-wave" if not flag else ("pipelined, one block/CU" if blocks == 2 else "pipelined, two blocks/CU")
        print(f"{Cin}->{Cout} {dims}: {what:26s} fwd {tf*1e3:8.1f} us  {fl/tf/1e9:6.1f} TF alg", flush=True)
