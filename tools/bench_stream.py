"""HBM-bound kernels of the hot path: achieved bandwidth against their algorithmic bytes (GPU box).

    python tools/bench_stream.py [--json out.json]

For every kernel and shape: average launch time (HIP events on the launch stream, 20 launches after 3 warm-ups),
algorithmic bytes per launch (each operand read / written once, fp32) and the resulting TB/s next to the
8 TB/s spec / 6.29 TB/s measured-copy peaks of MI355X_MICROARCH.md.  tools/pmc_hbm.py collects the PMC
FETCH_SIZE / WRITE_SIZE of the same launches."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rehrseg_amd import hip_backend as hb  # noqa: E402
from rehrseg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(2, 32, 128, 128, 128), (2, 64, 64, 64, 64), (2, 128, 32, 32, 32), (1, 64, 128, 64, 64), (1, 64, 128, 128, 128),
          (1, 128, 128, 32, 32), (1, 512, 128, 16, 16)]


def act_t(shape):
    return torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)


def timed(fn, reps=20):
    for _ in range(3):
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
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="substring filter on kernel names (PMC passes run one at a time)")
    args = ap.parse_args()
    rows = []

    def add(name, shape, nbytes, fn):
        if args.only and args.only not in name:
            return
        t = timed(fn)
        rows.append({"kernel": name, "shape": list(shape), "us": t * 1e6, "algorithmic_bytes": nbytes,
                     "TBps": nbytes / t / 1e12, "frac_of_8TBps": nbytes / t / 8e12, "frac_of_6p29": nbytes / t / 6.29e12})
        print(f"{name:28s} {str(shape):28s} {t * 1e6:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / t / 1e12:6.2f} TB/s "
              f"({nbytes / t / 8e12:5.1%} of 8 TB/s)", flush=True)

    for shape in SHAPES:
        N, C, D, H, W = shape
        S = D * H * W
        el = N * C * S
        x, dy = act_t(shape), act_t(shape)
        if C <= 320:
            stats = torch.zeros((N, C, 2), dtype=torch.float64, device=dev)
            stats[..., 0] = x.double().sum((2, 3, 4))
            stats[..., 1] = (x.double() ** 2).sum((2, 3, 4))
            g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
            y, mr = hb.instnorm_act_fwd(x, stats, g, b, 1e-5, 2, 0.01)
            add("instnorm_act_fwd", shape, 8 * el, lambda: hb.instnorm_act_fwd(x, stats, g, b, 1e-5, 2, 0.01))
            add("instnorm_act_bwd(2 passes)", shape, 20 * el, lambda: hb.instnorm_act_bwd(dy, x, mr, g, b, 2, 0.01))
        gate = torch.rand((N, C), device=dev)
        res = act_t(shape)
        y = hb.scale_res_act_fwd(x, gate, res, 1, 0.0)
        add("scale_res_act_fwd(+res)", shape, 12 * el, lambda: hb.scale_res_act_fwd(x, gate, res, 1, 0.0))
        add("scale_res_act_bwd(+dres)", shape, 20 * el, lambda: hb.scale_res_act_bwd(dy, y, x, gate, True, 1, 0.0))
        k = torch.randn((N, C), device=dev)
        add("add_channel_const", shape, 8 * el, lambda: hb.add_channel_const(dy, k))
        add("act_bwd", shape, 12 * el, lambda: hb.act_bwd(dy, y, 1, 0.0))
        add("channel_sum", shape, 4 * el, lambda: hb.channel_sum(dy))
        del x, dy, res, y
    # SegModel heads at cfg-3: upmix (interpolate + depth-tap sum) and the fused loss on the HR logits
    gsh = (2, 48, 128, 128, 128)
    gt = act_t(gsh)
    bias = torch.randn(16, device=dev)
    yup = hb.upmix_depth_fwd(gt, bias, 512, 16, 3, 1, 1, 0.0)
    dyup = act_t((2, 16, 512, 128, 128))
    add("upmix_depth_fwd", gsh, 4 * gt.numel() + 4 * yup.numel(), lambda: hb.upmix_depth_fwd(gt, bias, 512, 16, 3, 1, 1, 0.0))
    add("upmix_depth_bwd", gsh, 4 * gt.numel() + 8 * yup.numel(), lambda: hb.upmix_depth_bwd(dyup, yup, 128, 3, 1, 1, 0.0))
    add("channel_sum_actgrad", (2, 16, 512, 128, 128), 8 * yup.numel(), lambda: hb.channel_sum_actgrad(dyup, yup, 1, 0.0))
    del gt, yup, dyup
    lg = act_t((2, 2, 512, 128, 128))
    tgt = torch.randint(0, 2, (2, 512 * 128 * 128), device=dev).float()
    st = hb.seg_loss_fwd(lg, tgt, None)
    go = torch.ones(1, device=dev)
    add("seg_loss_fwd", (2, 2, 512, 128, 128), 4 * lg.numel() + 4 * tgt.numel(), lambda: hb.seg_loss_fwd(lg, tgt, None))
    add("seg_loss_bwd", (2, 2, 512, 128, 128), 8 * lg.numel() + 4 * tgt.numel(),
        lambda: hb.seg_loss_bwd(lg, tgt, None, st, 1.0, 1.0, 1e-5, False, go))
    # UASR head of the reference-shape step (32 x 4 slices of 96 x 96, K = 16 candidate pairs)
    om = act_t((32, 128, 1, 96, 96))
    ue = act_t((32, 64, 1, 96, 96))
    wu, bu = torch.randn(16, device=dev), torch.randn(1, device=dev)
    uo, uu = hb.uasr_mix_fwd(om, ue, wu, bu, 4)
    go_, gu_ = torch.randn_like(uo), torch.randn_like(uu)
    vs = 32 * 4 * 96 * 96   # voxel-slices
    add("uasr_mix_fwd", (32, 192, 1, 96, 96), vs * (12 * 16 + 12), lambda: hb.uasr_mix_fwd(om, ue, wu, bu, 4))
    add("uasr_mix_bwd", (32, 192, 1, 96, 96), vs * (24 * 16 + 24), lambda: hb.uasr_mix_bwd(om, ue, wu, bu, go_, gu_, 4))
    del om, ue, uo, uu, go_, gu_
    if args.json:
        json.dump({"peak_spec_TBps": 8.0, "peak_measured_copy_TBps": 6.29, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
