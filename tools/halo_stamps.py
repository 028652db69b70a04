"""Diagnostic: per-phase cycle counts of halo_conv_bf16_kernel (library built with -DHB_STAMPS; the stamps go to a
buffer of their own handed over in the descriptor's unused scratch pointer).  GPU box only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rehrseg_amd import hip_backend as hb, lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
N, Cin, Cout, D, H, W = (int(a) for a in (sys.argv[1:7] if len(sys.argv) > 6 else (2, 32, 32, 128, 128, 128)))
x = torch.randn(N, Cin, D, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
wp, Npad = ops._pack(w, 0, torch.bfloat16)
y = hb.new_act(N, Cout, D, H, W, like=x)
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
d = L.GatherGemmDesc()
taps = [ops.full_taps(3)] * 3
hb._gg_desc(d, x, None, Cin, (D, H, W), Cin, (D, H, W), (1, 1, 1), (-1, -1, -1), taps, 3, 3, wp, Npad, y, (D, H, W), Cout,
            (1, 1, 1), (0, 0, 0), None, 0, 0.0, None, 0, ops.choose_tile((D, H, W)))
d.wino_ws, d.wino_ws_bytes = C.c_void_p(dbg.data_ptr()), 64
for _ in range(3):
    L.check(L.load().rehr_gather_gemm_bf16(C.byref(d), hb._stream()), "gg")
torch.cuda.synchronize()
print(dbg.cpu().tolist()); f, s, e, st, items = (int(v) for v in dbg[:5].cpu())
print(f"items {items}: per item cycles (s_memtime ticks): fetch-issue {f / items:.0f}  sweep {s / items:.0f}  "
      f"epilogue {e / items:.0f}  barrier+stage {st / items:.0f}")
