"""Conv + InstanceNorm + LeakyReLU blocks at the cfg-3 (nnU-Net) stage shapes: forward and backward (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(32, 32, (128, 128, 128)), (64, 32, (128, 128, 128)), (64, 64, (64, 64, 64)), (128, 128, (32, 32, 32)),
          (256, 256, (16, 16, 16)), (320, 320, (8, 8, 8))]


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for Cin, Cout, dims in SHAPES:
    x = torch.randn(2, Cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02).requires_grad_(True)
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    ga = torch.ones(Cout, device=dev, requires_grad=True)
    be = torch.zeros(Cout, device=dev, requires_grad=True)
    fwd = lambda: ops.fused_conv3d(x, w, b, 1, 1, inorm=(ga, be), act=ops.ACT_LRELU, slope=0.01)
    y = fwd()
    g = torch.randn_like(y)
    def both():
        y = fwd()
        torch.autograd.grad(y, [x, w, b, ga, be], g)
    tf = timeit(fwd)
    tb = timeit(both) - tf
    flops = 2.0 * 2 * dims[0] * dims[1] * dims[2] * 27 * Cin * Cout
    print(f"{Cin:4d}->{Cout:4d} {dims}: fwd {tf*1e3:7.3f} ms {flops/tf/1e12:6.1f} TF | bwd {tb*1e3:7.3f} ms {2*flops/tb/1e12:6.1f} TF", flush=True)
