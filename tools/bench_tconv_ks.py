"""kernel == stride transposed convolutions of the nnU-Net plans: fused-phase kernel (tconv_ks.hip) against the generic
one-block-per-(tile, phase) grid, fp32 and bf16.  python tools/bench_tconv_ks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import hip_backend as hb, ops
dev = torch.device("cuda:0")

def t(f, n=10):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

print("lib:", os.environ.get("REHRSEG_HIP_LIB", "in-tree"))
for (N, Cin, Cout, dims, s) in [(2, 64, 32, (64, 64, 64), (2, 2, 2)), (2, 128, 64, (32, 32, 32), (2, 2, 2)),
                                (1, 64, 32, (160, 80, 80), (1, 2, 2)), (1, 128, 64, (160, 40, 40), (1, 2, 2))]:
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(N, Cin, *dims, device=dev).to(dt).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(Cin, Cout, *s, device=dev) * 0.05
        b = torch.zeros(Cout, device=dev)
        cfg = ops.ConvCfg(s, (0, 0, 0), True)
        wp, Npad = ops._pack(w, 1, dt)
        es = 4 if dt == torch.float32 else 2
        vox = N * dims[0] * dims[1] * dims[2]
        nb = vox * Cin * es + vox * s[0] * s[1] * s[2] * Cout * es
        line = f"{N}x{Cin}->{Cout} {dims} s{s} {str(dt)[6:]:9s}"
        for flag in (True, False):
            hb.USE_TCONV_KS = flag
            ms = t(lambda: ops.conv_forward(x, None, w, b, cfg, ops.ACT_NONE, 0.0, 0))
            line += f"  {'fused' if flag else 'generic'} {ms * 1e3:7.1f} us ({nb / ms / 1e9:5.2f} TB/s incl. pack)"
        print(line, flush=True)
