"""sr_head.2 (Conv3d 16->2, 5x5x5) on the bf16 matrix cores vs the fp32 VALU kernels (GPU box): check + timing.

    python tools/bench_thin5.py [--only fwd]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from rehrseg_amd import hip_backend as hb, ops  # noqa: E402

dev = torch.device("cuda:0")


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
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp32"))
    args = ap.parse_args()
    DT = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    for (N, D, H, W) in ((1, 12, 10, 32), (2, 40, 30, 64), (1, 23, 9, 96), (2, 512, 128, 128), (1, 640, 160, 160))[:4 if args.dtype == "fp32" else 5]:
        x = torch.randn(N, 16, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
        xb = x.to(DT)
        w = torch.randn(2, 16, 5, 5, 5, device=dev) * 0.05
        b = torch.randn(2, device=dev)
        small = D * H * W < 2 ** 18
        flop = 2.0 * N * D * H * W * 16 * 2 * 125
        if args.only in (None, "fwd"):
            y = hb.thin5_fwd(xb, w, b)
            if small:
                ref = F.conv3d(xb.float(), w.to(DT).float(), b, padding=2)
                err = ((y - ref).abs().max() / ref.abs().max()).item()
            else:  # against the fp32 VALU kernel on the same rounded operands
                ref, _ = ops.conv_forward(xb.float(), None, w.to(DT).float(), b, ops.ConvCfg((1, 1, 1), (2, 2, 2)), 0, 0.0, 0)
                err = ((y - ref).abs().max() / ref.abs().max()).item()
            t = timed(lambda: hb.thin5_fwd(xb, w, b))
            t0 = timed(lambda: ops.conv_forward(x, None, w, b, ops.ConvCfg((1, 1, 1), (2, 2, 2)), 0, 0.0, 0))
            print(f"fwd  {(N, D, H, W)}: max rel err {err:.2e}  mfma {t * 1e6:8.1f} us ({flop / t / 1e12:6.1f} TF alg)  "
                  f"valu fp32 {t0 * 1e6:8.1f} us", flush=True)
        dy = torch.randn(N, 2, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
        dyb = dy.to(DT).float()
        wb = w.to(DT).float()
        cfg = ops.ConvCfg((1, 1, 1), (2, 2, 2))
        if args.only in (None, "dgrad"):
            dx = hb.thin5_dgrad(dy, w, DT).float()
            if small:
                ref = torch.nn.grad.conv3d_input(x.shape, wb, dyb, padding=2)
            else:
                ref, _ = ops.conv_dgrad(dyb, wb, (D, H, W), 16, 0, cfg)
            err = ((dx - ref).abs().max() / ref.abs().max()).item()   # includes the bf16 rounding of the result
            t = timed(lambda: hb.thin5_dgrad(dy, w, DT))
            t0 = timed(lambda: ops.conv_dgrad(dy, w, (D, H, W), 16, 0, cfg))
            print(f"dgrad{(N, D, H, W)}: max rel err {err:.2e}  mfma {t * 1e6:8.1f} us ({flop / t / 1e12:6.1f} TF alg)  "
                  f"valu fp32 {t0 * 1e6:8.1f} us", flush=True)
        if args.only in (None, "wgrad"):
            dw, _ = hb.thin5_wgrad(xb, w, dy)
            if small:
                ref = torch.nn.grad.conv3d_weight(xb.float(), w.shape, dyb, padding=2)
            else:
                ref, _ = ops.conv_wgrad(dyb, xb.float(), None, w, cfg, False)
            err = ((dw - ref).abs().max() / ref.abs().max()).item()
            t = timed(lambda: hb.thin5_wgrad(xb, w, dy))
            t0 = timed(lambda: ops.conv_wgrad(dy, x, None, w, cfg, False))
            print(f"wgrad{(N, D, H, W)}: max rel err {err:.2e}  mfma {t * 1e6:8.1f} us ({flop / t / 1e12:6.1f} TF alg)  "
                  f"valu fp32 {t0 * 1e6:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
