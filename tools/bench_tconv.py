"""FLAVR decoder transposed convs (3,4,4)/(1,2,2): forward + input gradient timing (F(2x2,2x2) kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rehrseg_amd import ops
dev = torch.device("cuda:0")
def t(f, n=5):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Cin, Cout, dims) in [(512, 128, (128, 16, 16)), (256, 64, (128, 32, 32)), (128, 64, (128, 64, 64))]:
    x = torch.randn(1, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cin, Cout, 3, 4, 4, device=dev) * 0.02
    b = torch.zeros(Cout, device=dev)
    cfg = ops.ConvCfg((1, 2, 2), (1, 1, 1), True)
    res = {}
    for mode in ("0",):
        y, _ = ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.2, 0)
        dy = torch.ones_like(y) if "dy" not in res else res["dy"]
        res["dy"] = dy
        dx, _ = ops.conv_dgrad(dy, w, dims, Cin, 0, cfg)
        tf = t(lambda: ops.conv_forward(x, None, w, b, cfg, ops.ACT_LRELU, 0.2, 0))
        tb = t(lambda: ops.conv_dgrad(dy, w, dims, Cin, 0, cfg))
        if mode == "0":
            res["y"], res["dx"] = y.clone(), dx.clone()
            e1 = e2 = 0.0
        else:
            e1 = ((y - res["y"]).abs().max() / res["y"].abs().max()).item()
            e2 = ((dx - res["dx"]).abs().max() / res["dx"].abs().max()).item()
        print(f"{Cin}->{Cout} {dims}: fwd {tf:6.3f} ms  dgrad {tb:6.3f} ms  diff {e1:.1e} {e2:.1e}", flush=True)
