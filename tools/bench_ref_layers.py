"""Forward / input-gradient timing of the reference-shape (32 x 4 slices of 96x96 crops) encoder layers (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rehrseg_amd import ops, hip_backend
dev = torch.device("cuda:0")
if len(sys.argv) > 1:   # tiles per block of the flattened-tile kernel: 0 = the library's pick, 1 = 32, 2 = 64
    hip_backend.WINO_FLAT8_TILES = int(sys.argv[1])
    print("WINO_FLAT8_TILES =", hip_backend.WINO_FLAT8_TILES)
def t(f, n=5):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for Cin, Cout, hw in [(64, 64, 48), (128, 128, 24), (256, 256, 12), (512, 512, 12), (512, 256, 12)]:
    N, D = 32, 4
    x = torch.randn(N, Cin, D, hw, hw, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.02
    cfg = ops.ConvCfg((1, 1, 1), (1, 1, 1), False)
    y, _ = ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0)
    dy = torch.randn_like(y)
    # reachable taps only (depth 4 with 3 taps: 10 of 12 (slice, tap) pairs; H, W borders likewise)
    rd = 10.0 / 12.0; rh = (3 * hw - 2) / (3.0 * hw)
    fl = 2.0 * N * D * hw * hw * 27 * Cin * Cout * rd * rh * rh
    tf = t(lambda: ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 0))
    tb = t(lambda: ops.conv_dgrad(dy, w, (D, hw, hw), Cin, 0, cfg))
    tw = t(lambda: ops.conv_wgrad(dy, x, None, w, cfg, True))
    print(f"{Cin}->{Cout} {hw}x{hw}: fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.1f} TF | dgrad {tb*1e3:7.1f} us | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.1f} TF", flush=True)
