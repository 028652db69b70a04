"""north_star's literal acceptance test: the HIP path against the CPU reference path (oracle/, fp32 on the
host) on identical synthetic inputs and identical weights at BASELINE.json's full sizes --

  cfg-2  UNet_3D_3D(1,'unet_18',128,4) on 1x1x128^3   (oracle/flavr_oracle.py, pinned by reference fixtures)
  cfg-3  SegModel isotropic 3d_fullres plan on 2x1x128^3 (oracle/segmodel_oracle.py: in-reference parts pinned
         by G8 fixtures, nnU-Net bases "parity unpinned")

Asserted: forward max-rel <= 1e-3 (observed ~1e-6), loss rel <= 1e-4, label maps (argmax) exact wherever the
logit margin exceeds 1e-3 of the logit scale (mismatches on near-ties are counted and printed), and every
parameter gradient within 1e-3 l2-relative of the oracle's fp32 gradient OR within 3x the distance between
the oracle's own fp32 and fp64 gradients for that tensor -- the legitimate conditioning yardstick: where the
reference's fp32 CPU run itself is only defined to 2e-3 (deep, cancelling gradients at random initialisation),
no fp32 implementation can be closer to it than that.  The full table is printed and written to
gpurun_out/parity_<cfg>.json; DESIGN.md section 5 quotes it.
One oracle step takes ~30 s in fp32; the fp64 legs take 110 s (cfg-2) and 350 s (cfg-3, one sample at a time) on the GPU
box's 128 host threads, so by default the yardstick is read from tests/golden/conditioning_cfg{2,3}.json -- the
fp32-vs-fp64 distances this very test measured there (same deterministic inputs and weights; REHR_PARITY_FP64=1
recomputes them live and is how the fixtures and profiles/r02_parity_cfg*.json were produced).
"""
import json
import os
import threading
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIVE_FP64 = os.environ.get("REHR_PARITY_FP64", "0") == "1"


def _dtypes():
    return (torch.float32, torch.float64) if LIVE_FP64 else (torch.float32,)


def _conditioning(tag):
    return json.load(open(os.path.join(ROOT, "tests", "golden", f"conditioning_{tag}.json")))


class heartbeat:
    """The fp64 oracle leg runs for minutes on the host: keep gpurun_out/ changing so the run is not taken for hung."""

    def __init__(self, tag):
        self.tag, self.stop = tag, threading.Event()

    def __enter__(self):
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        t0 = time.time()

        def beat():
            while not self.stop.wait(45.0):
                with open(os.path.join(out, f"heartbeat_{self.tag}.txt"), "w") as f:
                    f.write(f"{self.tag}: oracle running, {time.time() - t0:.0f} s\n")
        self.th = threading.Thread(target=beat, daemon=True)
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.th.join()


def _l2rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def _gradient_table(hip_grads, g32, g64, skip=(), cond=None):
    """g64: the oracle's fp64 gradients (live run) or None, in which case `cond` holds the recorded distances."""
    rows, bad = [], []
    for k, ref in g32.items():
        if k in skip or ref is None:
            continue
        n = float(ref.double().norm())
        if n == 0.0:
            continue
        d_hip = _l2rel(hip_grads[k].cpu(), ref)
        if g64 is not None:
            d_cond = _l2rel(ref, g64[k])          # the reference path's own fp32 rounding distance
            d_hip64 = _l2rel(hip_grads[k].cpu(), g64[k])
        else:
            d_cond, d_hip64 = float(cond[k]), None
        rows.append({"param": k, "hip_vs_cpu_fp32": d_hip, "cpu_fp32_vs_fp64": d_cond, "hip_vs_cpu_fp64": d_hip64})
        if d_hip > 1e-3 and d_hip > 3.0 * d_cond:
            bad.append(rows[-1])
    return rows, bad


def _report(tag, summary, rows):
    over = [r for r in rows if r["hip_vs_cpu_fp32"] > 1e-3]
    worst = max(rows, key=lambda r: r["hip_vs_cpu_fp32"])
    summary.update(n_gradients=len(rows), n_over_1e3=len(over), worst=worst,
                   median_hip_vs_cpu_fp32=sorted(r["hip_vs_cpu_fp32"] for r in rows)[len(rows) // 2],
                   max_hip_vs_cpu_fp64=(max(r["hip_vs_cpu_fp64"] for r in rows) if rows[0]["hip_vs_cpu_fp64"] is not None else None),
                   max_cpu_fp32_vs_fp64=max(r["cpu_fp32_vs_fp64"] for r in rows))
    print(f"[{tag}] " + json.dumps(summary))
    for r in sorted(rows, key=lambda r: -r["hip_vs_cpu_fp32"])[:12]:
        h64 = "recorded" if r["hip_vs_cpu_fp64"] is None else f"{r['hip_vs_cpu_fp64']:.2e}"
        print(f"[{tag}]   {r['param']:55s} hip-vs-cpu32 {r['hip_vs_cpu_fp32']:.2e}   cpu32-vs-cpu64 "
              f"{r['cpu_fp32_vs_fp64']:.2e}   hip-vs-cpu64 {h64}")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"parity_{tag}.json"), "w") as f:
        json.dump({"summary": summary, "gradients": rows}, f, indent=1)


def test_cfg2_flavr_128cube_against_cpu_reference_path():
    from oracle import flavr_oracle as fo
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    dev = torch.device("cuda:0")
    m = UNet_3D_3D(1, "unet_18", 128, 4)
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(dev)
    x = det_input("cfg2.x", (1, 1, 128, 128, 128), "rand")
    tgt = det_input("cfg2.t", (1, 1, 4, 128, 128), "rand")
    xin = x.clone().to(dev)
    out = m(xin)
    loss = (out - tgt.to(dev)).abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    hip = {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}

    runs = {}
    for dt in _dtypes():
        with heartbeat("cfg2"):
            t0 = time.time()
            osd = {k: v.detach().clone().to(dt).requires_grad_() for k, v in sd.items()}
            xr = x.to(dt).clone()
            r = fo.unet_3d_3d(osd, xr, 1, 128, 4)
            rl = (r - tgt.to(dt)).abs().mean()
            rl.backward()
            runs[dt] = (r.detach(), float(rl.detach()), {k: v.grad for k, v in osd.items()}, xr)
            print(f"[cfg2] oracle {dt} step: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    r32, l32, g32, x32 = runs[torch.float32]
    fwd = float((out.detach().cpu() - r32).abs().max() / r32.abs().max())
    if LIVE_FP64:
        rows, bad = _gradient_table(hip, g32, runs[torch.float64][2])
        extra = {"loss_cpu_fp64": runs[torch.float64][1], "yardstick": "live fp64 run",
                 "fwd_cpu_fp32_vs_fp64": float((r32.double() - runs[torch.float64][0]).abs().max() /
                                               runs[torch.float64][0].abs().max())}
    else:
        cj = _conditioning("cfg2")
        rows, bad = _gradient_table(hip, g32, None, cond=cj["cpu_fp32_vs_fp64"])
        extra = {"loss_cpu_fp64": cj["loss_cpu_fp64"], "yardstick": "tests/golden/conditioning_cfg2.json"}
    _report("cfg2", dict({"fwd_max_rel": fwd, "loss_hip": float(loss), "loss_cpu_fp32": l32}, **extra), rows)
    assert fwd <= 1e-3
    assert abs(float(loss) - l32) <= 1e-4 * abs(l32)
    assert torch.allclose(xin.cpu(), x32, atol=1e-6)   # the in-place mean subtraction of the caller's tensor (:181)
    assert not bad, bad


def test_cfg3_segmodel_128cube_against_cpu_reference_path():
    from oracle import aux_oracle as ao
    from oracle import segmodel_oracle as so
    from oracle.detinit import det_input
    from rehrseg_amd.utils.seg_utils import _build_loss
    from test_segmodel_cpu import build
    dev = torch.device("cuda:0")
    m, sd = build(so.ISO_PLAN, dev)
    x = det_input("cfg3.x", (2, 1, 128, 128, 128), "randn")
    lab_lr = det_input("cfg3.lab_lr", (2, 1, 128, 128, 128), "randint2")
    lab_hr = det_input("cfg3.lab_hr", (2, 1, 512, 128, 128), "randint2")
    crit = _build_loss()
    out, out_up = m(x.to(dev))
    loss = crit(out, lab_lr.to(dev)) + crit(out_up, lab_hr.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    hip = {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}
    out_c, up_c, loss_val = out.detach().cpu(), out_up.detach().cpu(), float(loss)
    del out, out_up, loss
    torch.cuda.empty_cache()

    runs = {}
    N = x.shape[0]
    for dt in _dtypes():
        with heartbeat("cfg3"):
            t0 = time.time()
            osd = {k: v.detach().clone().to(dt).requires_grad_() for k, v in sd.items() if k in so.segmodel_shapes(so.ISO_PLAN)}
            if dt == torch.float32:   # the reference CPU path as it runs: the whole batch at once
                r_out, r_up = so.seg_model(osd, x.to(dt), so.ISO_PLAN)
                rl = ao.dc_and_weighted_ce(r_out, lab_lr.to(dt)) + ao.dc_and_weighted_ce(r_up, lab_hr.to(dt))
                rl.backward()
                runs[dt] = (r_out.detach(), r_up.detach(), float(rl.detach()), {k: v.grad for k, v in osd.items()})
                del r_out, r_up, rl
            else:
                # fp64 yardstick, one sample at a time: ATen's double-precision Conv3d on the CPU unfolds its input
                # (27 x C_in columns per voxel: 116 GB for sr_head.0 on the 2x32x512x128x128 tensor), and nothing in
                # the model or the loss couples samples (InstanceNorm and soft Dice are per sample, CE is a mean),
                # so the batch loss is the mean of the per-sample losses and the gradients add up.
                tot = 0.0
                for b in range(N):
                    o_b, u_b = so.seg_model(osd, x[b:b + 1].to(dt), so.ISO_PLAN)
                    l_b = (ao.dc_and_weighted_ce(o_b, lab_lr[b:b + 1].to(dt)) +
                           ao.dc_and_weighted_ce(u_b, lab_hr[b:b + 1].to(dt))) / N
                    l_b.backward()
                    tot += float(l_b.detach())
                    del o_b, u_b, l_b
                runs[dt] = (None, None, tot, {k: v.grad for k, v in osd.items()})
            del osd
            print(f"[cfg3] oracle {dt} step: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    r_out, r_up, l32, g32 = runs[torch.float32]
    fwd = max(float((a - b).abs().max() / b.abs().max()) for a, b in ((out_c, r_out), (up_c, r_up)))
    mism = {}
    for name, a, b in (("lr", out_c, r_out), ("hr", up_c, r_up)):
        la, lb = a.argmax(1), b.argmax(1)
        clear = (b[:, 0] - b[:, 1]).abs() > 1e-3 * float(b.abs().max())
        assert bool((la == lb)[clear].all()), name            # label maps: bit-exact outside the 1e-3 margin
        mism[name] = [int((la != lb).sum()), la.numel()]
    # conv biases sit in front of InstanceNorm: their gradient is identically 0, what is left is rounding noise
    skip = [k for k in g32 if k.endswith("conv.bias")]
    if LIVE_FP64:
        rows, bad = _gradient_table(hip, g32, runs[torch.float64][3], skip)
        extra = {"loss_cpu_fp64": runs[torch.float64][2], "yardstick": "live fp64 run"}
    else:
        cj = _conditioning("cfg3")
        rows, bad = _gradient_table(hip, g32, None, skip, cond=cj["cpu_fp32_vs_fp64"])
        extra = {"loss_cpu_fp64": cj["loss_cpu_fp64"], "yardstick": "tests/golden/conditioning_cfg3.json"}
    _report("cfg3", dict({"fwd_max_rel": fwd, "loss_hip": loss_val, "loss_cpu_fp32": l32,
                          "argmax_mismatches_on_near_ties": mism}, **extra), rows)
    assert fwd <= 1e-3
    assert abs(loss_val - l32) <= 1e-4 * abs(l32)
    assert not bad, bad
