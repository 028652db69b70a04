"""Mixed-precision conv layers one at a time (GPU box): forward / input gradient / weight gradient TFLOP/s of the
bf16 kernels on the SegModel / FLAVR layer shapes at 128^3 - 160^3.

    python tools/bench_bf16_layers.py [--only fwd] [--shape i]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rehrseg_amd import hip_backend, ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
SHAPES = [  # N, Cin, Cout, D, H, W, K
    (2, 32, 32, 128, 128, 128, (3, 3, 3)), (2, 64, 32, 128, 128, 128, (3, 3, 3)), (2, 64, 64, 64, 64, 64, (3, 3, 3)),
    (2, 128, 128, 32, 32, 32, (3, 3, 3)), (2, 256, 256, 16, 16, 16, (3, 3, 3)), (2, 320, 320, 8, 8, 8, (3, 3, 3)),
    (1, 32, 32, 160, 160, 160, (1, 3, 3)), (1, 64, 64, 160, 80, 80, (1, 3, 3)), (1, 64, 64, 128, 64, 64, (3, 3, 3)),
]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--shape", type=int, default=None)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    for si, (N, Cin, Cout, D, H, W, K) in enumerate(SHAPES):
        if args.shape is not None and si != args.shape:
            continue
        pad = tuple(k // 2 for k in K)
        x = torch.randn(N, Cin, D, H, W, device=dev).to(BF).contiguous(memory_format=torch.channels_last_3d)
        dz = torch.randn(N, Cout, D, H, W, device=dev).to(BF).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn((Cout, Cin) + K, device=dev) * 0.05
        cfg = ops.ConvCfg((1, 1, 1), pad)
        flop = 2.0 * N * D * H * W * Cin * Cout * K[0] * K[1] * K[2]
        line = f"{str((N, Cin, Cout, D, H, W, K)):44s}"
        fwd = lambda: ops.conv_forward(x, None, w, None, cfg, 0, 0.0, 2)  # noqa: E731
        for name, fn in (("fwd", fwd), ("fwd_gather", fwd),
                         ("dgrad", lambda: ops.conv_dgrad(dz, w, (D, H, W), Cin, 0, cfg)),
                         ("wgrad", lambda: ops.conv_wgrad(dz, x, None, w, cfg, False))):
            if args.only and args.only != name:
                continue
            hip_backend.USE_HALO_BF16 = name != "fwd_gather"
            t = timed(fn, args.reps)
            hip_backend.USE_HALO_BF16 = True
            line += f"  {name} {t * 1e6:8.1f} us {flop / t / 1e12:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
