import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops, hip_backend
dev = torch.device("cuda:0")
# feature_fuse at cfg-2: (n_inputs,3,3) conv on (1,64,128,128,128) -> (1,64,1,128,128)
for D in (8, 128):
    x = torch.randn(1, 64, D, 128, 128, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.randn(64, 64, D, 3, 3, device=dev) * 0.02).requires_grad_(True)
    b = torch.zeros(64, device=dev, requires_grad=True)
    f = lambda: ops.fused_conv3d(x, w, b, 1, (0, 1, 1), act=ops.ACT_LRELU, slope=0.2)
    y = f(); g = torch.randn_like(y)
    res = {}
    for mode in ("0", "1"):
        hip_backend.USE_WINOGRAD = mode == "1"
        hip_backend.USE_WINOGRAD_WGRAD = mode != "0"
        y = f()
        gr = torch.autograd.grad(y, [x, w, b], g)
        res[mode] = [y] + list(gr)
        for _ in range(2):
            y = f(); torch.autograd.grad(y, [x, w, b], g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = f(); torch.autograd.grad(y, [x, w, b], g)
        e1.record(); torch.cuda.synchronize()
        print(f"D={D} winograd={mode}: fwd+bwd {e0.elapsed_time(e1)/5:.3f} ms", flush=True)
    for nm, a, c in zip(("y", "dx", "dw", "db"), res["0"], res["1"]):
        print("   ", nm, ((a - c).abs().max() / a.abs().max()).item())
