"""BASELINE.json configs[3] and configs[4] as `bench.py --workload cfg4 | cfg5` runs them, under a checker:

  cfg-4  joint stage-2 step (frozen FLAVR teacher over D-1 windows + SegModel student on the distillation-compatible
         anisotropic plan + Distiller + uncertainty-weighted CE / DC+CE + SGD) on ONE 1x1x128^3 LR patch, fp32,
         against the CPU reference path (oracle/, fp32, identical deterministic weights and inputs): loss, LR / HR
         label maps, teacher features, and every parameter gradient with the same rule as
         tests/test_full_size_oracle_gpu.py (<= 1e-3 l2-rel, or <= 3x the oracle's own fp32-vs-fp64 distance recorded in
         tests/golden/conditioning_cfg4.json by a REHR_PARITY_FP64=1 run of this very test).
  cfg-5  the same step at 1x1x160^3 under ops.mixed_precision() against the fp32 HIP step on the same inputs (the HIP
         fp32 path is what the cfg-4 test and tests/test_full_size_oracle_gpu.py tie to the CPU path): finite, loss
         within 1e-3, teacher features within 1e-2 l2-rel, logits within 3e-2 l2-rel, LR / HR label agreement reported
         and >= 99 % / exact wherever the fp32 logit margin exceeds 5 % of the logit scale.  Gradients: every tensor's
         bf16-vs-fp32 distance is reported; the bar is by depth, as calibrated by the full-depth-plan test of
         tests/test_mixed_steps_gpu.py (there the bf16-emulating fp64 oracle shows what rounding ALONE does to this
         randomly initialised 22-layer InstanceNorm stack: percent at the heads, tens of percent at the bottom stages,
         whose gradients are small residuals of cancelling sums): heads / last decoder stage <= 2e-2, everything
         <= 0.6 l2-rel with cosine similarity >= 0.8.

Reference: train_all.py:85-112 (get_intermediate_features), :519-556 (the loop).
"""
import itertools
import json
import os
import time

import pytest
import torch

from test_full_size_oracle_gpu import LIVE_FP64, ROOT, _gradient_table, _l2rel, _report, heartbeat

pytestmark = pytest.mark.gpu


def _build(dev, size):
    from oracle import segmodel_oracle as so
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    from rehrseg_amd.models.seg_model import Distiller
    from test_segmodel_cpu import build
    teacher = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True)
    tsd = {k: det_tensor(k, tuple(v.shape)) for k, v in teacher.state_dict().items()}
    teacher.load_state_dict(tsd)
    teacher = teacher.to(dev).eval()
    for q in teacher.parameters():
        q.requires_grad_(False)
    student, ssd = build(so.ANISO_PLAN, dev)
    dist = Distiller(64, 64, 0.0, 1.0, 1.0)
    dsd = {k: det_tensor(k, tuple(v.shape)) for k, v in dist.state_dict().items()}
    dist.load_state_dict(dsd)
    dist = dist.to(dev)
    tag = f"joint{size}"
    img = det_input(tag + ".img", (1, 1, size, size, size), "randn")
    lab_lr = det_input(tag + ".lab_lr", (1, 1, size, size, size), "randint2")
    lab_hr = det_input(tag + ".lab_hr", (1, 1, 4 * size, size, size), "randint2")
    unc = 1.0 - torch.floor(det_input(tag + ".unc", (1, 1, size, size, size), "rand") * 256) / 255.0 * 0.99   # train_set.py:148
    return teacher, tsd, student, ssd, dist, dsd, img, lab_lr, lab_hr, unc


def _hip_step(teacher, student, dist, img, lab_lr, lab_hr, unc, dev, mixed):
    """One train_segsr_step (lr = 0: the parameters stay put, the gradients stay in .grad) plus the quantities the
    checks read: teacher level-1 features and the student's logits on the z-scored image."""
    from rehrseg_amd import ops
    from rehrseg_amd.train_steps import get_intermediate_features, train_segsr_step
    from rehrseg_amd.utils import seg_utils as su
    opt = torch.optim.SGD(itertools.chain(student.parameters(), dist.parameters()), lr=0.0)
    for p in itertools.chain(student.parameters(), dist.parameters()):
        p.grad = None
    with ops.mixed_precision(mixed):
        with torch.no_grad():
            imz = img.clone().to(dev)
            tf = get_intermediate_features(teacher, imz, lab_lr.to(dev), dev, levels=(1,))[1].float().cpu()
            s_lr, s_sr = student(imz)                            # imz was z-scored in place by the teacher pass
            s_lr, s_sr = s_lr.float().cpu(), s_sr.float().cpu()
        loss = train_segsr_step(student, teacher, dist, opt, img.clone().to(dev), lab_lr.to(dev), lab_hr.to(dev),
                                unc.to(dev), su._build_loss(False, weight_dice=0), su._build_loss(False, weight_dice=1))
    torch.cuda.synchronize()
    grads = {"student." + k: p.grad.detach().float().cpu() for k, p in student.named_parameters() if p.grad is not None}
    grads.update({"distiller." + k: p.grad.detach().float().cpu() for k, p in dist.named_parameters() if p.grad is not None})
    return float(loss), tf, s_lr, s_sr, grads


def test_cfg4_joint_step_128cube_against_cpu_reference_path():
    from oracle import aux_oracle as ao
    from oracle import segmodel_oracle as so
    from test_segmodel_cpu import canonical
    dev = torch.device("cuda:0")
    teacher, tsd, student, ssd, dist, dsd, img, lab_lr, lab_hr, unc = _build(dev, 128)
    loss_hip, tf_hip, lr_hip, sr_hip, g_hip = _hip_step(teacher, student, dist, img, lab_lr, lab_hr, unc, dev, False)
    g_hip = {("distiller." + k[10:] if k.startswith("distiller.") else "student." + canonical(k[8:])): v for k, v in g_hip.items()}
    torch.cuda.empty_cache()

    runs = {}
    for dt in ((torch.float32, torch.float64) if LIVE_FP64 else (torch.float32,)):
        with heartbeat("cfg4"):
            t0 = time.time()
            img_o = img.clone().to(dt)
            with torch.no_grad():
                tf = ao.teacher_features({k: v.to(dt) for k, v in tsd.items()}, img_o, lab_lr.to(dt), upto=1)[1]
            t_teacher = time.time() - t0
            o = {k: v.detach().to(dt).clone().requires_grad_() for k, v in ssd.items() if k in so.segmodel_shapes(so.ANISO_PLAN)}
            dw = dsd["distill.weight"].to(dt).clone().requires_grad_()
            db = dsd["distill.bias"].to(dt).clone().requires_grad_()
            s_lr, s_sr, sk = so.seg_model(o, img_o, so.ANISO_PLAN, return_features=True)
            rl = ao.dc_and_weighted_ce(s_lr, lab_lr.to(dt), unc.to(dt), weight_dice=0.0) + \
                ao.dc_and_weighted_ce(s_sr, lab_hr.to(dt), None) + ao.distiller_loss(dw, db, sk[1], tf, 0.0, 1.0, 1.0)
            rl.backward()
            g = {"student." + k: v.grad for k, v in o.items()}
            g.update({"distiller.distill.weight": dw.grad, "distiller.distill.bias": db.grad})
            runs[dt] = (float(rl.detach()), tf.detach(), s_lr.detach(), s_sr.detach(), g)
            del o, s_lr, s_sr, sk, rl
            print(f"[cfg4] oracle {dt} step: teacher {t_teacher:.1f} s + student {time.time() - t0 - t_teacher:.1f} s "
                  f"on {torch.get_num_threads()} threads")
    l32, tf32, lr32, sr32, g32 = runs[torch.float32]
    fwd = max(float((a - b).abs().max() / b.abs().max()) for a, b in ((lr_hip, lr32), (sr_hip, sr32)))
    tf_err = float((tf_hip - tf32).abs().max() / tf32.abs().max())
    mism = {}
    for name, a, b in (("lr", lr_hip, lr32), ("hr", sr_hip, sr32)):
        la, lb = a.argmax(1), b.argmax(1)
        clear = (b[:, 0] - b[:, 1]).abs() > 1e-3 * float(b.abs().max())
        assert bool((la == lb)[clear].all()), name            # label maps: bit-exact outside the 1e-3 margin
        mism[name] = [int((la != lb).sum()), la.numel()]
    skip = [k for k in g32 if k.endswith("conv.bias")]        # in front of InstanceNorm: identically zero gradient
    cpath = os.path.join(ROOT, "tests", "golden", "conditioning_cfg4.json")
    if LIVE_FP64:
        rows, bad = _gradient_table(g_hip, g32, runs[torch.float64][4], skip)
        extra = {"loss_cpu_fp64": runs[torch.float64][0], "yardstick": "live fp64 run"}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "conditioning_cfg4.json"), "w") as f:
            json.dump({"loss_cpu_fp64": runs[torch.float64][0],
                       "cpu_fp32_vs_fp64": {r["param"]: r["cpu_fp32_vs_fp64"] for r in rows}}, f, indent=1)
    else:
        cj = json.load(open(cpath))
        rows, bad = _gradient_table(g_hip, g32, None, skip, cond=cj["cpu_fp32_vs_fp64"])
        extra = {"loss_cpu_fp64": cj["loss_cpu_fp64"], "yardstick": "tests/golden/conditioning_cfg4.json"}
    _report("cfg4", dict({"fwd_max_rel": fwd, "teacher_features_max_rel": tf_err, "loss_hip": loss_hip,
                          "loss_cpu_fp32": l32, "argmax_mismatches_on_near_ties": mism}, **extra), rows)
    assert fwd <= 1e-3 and tf_err <= 1e-3
    assert abs(loss_hip - l32) <= 1e-4 * abs(l32)
    assert not bad, bad


def test_cfg5_joint_step_160cube_bf16_against_the_fp32_step():
    dev = torch.device("cuda:0")
    teacher, tsd, student, ssd, dist, dsd, img, lab_lr, lab_hr, unc = _build(dev, 160)
    l32, tf32, lr32, sr32, g32 = _hip_step(teacher, student, dist, img, lab_lr, lab_hr, unc, dev, False)
    l16, tf16, lr16, sr16, g16 = _hip_step(teacher, student, dist, img, lab_lr, lab_hr, unc, dev, True)
    assert all(torch.isfinite(t).all() for t in (tf16, lr16, sr16)) and all(torch.isfinite(v).all() for v in g16.values())
    summary = {"loss_fp32": l32, "loss_bf16": l16, "loss_rel": abs(l16 - l32) / abs(l32),
               "teacher_features_l2rel": _l2rel(tf16, tf32)}
    for name, a, b in (("lr", lr16, lr32), ("hr", sr16, sr32)):
        la, lb = a.argmax(1), b.argmax(1)
        margin = (b[:, 0] - b[:, 1]).abs()
        clear = margin > 5e-2 * float(b.abs().max())
        summary[f"argmax_agreement_{name}"] = float((la == lb).double().mean())
        summary[f"argmax_disagreements_{name}"] = [int((la != lb).sum()), la.numel()]
        summary[f"clear_voxels_{name}"] = float(clear.double().mean())
        summary[f"logits_l2rel_{name}"] = _l2rel(a, b)
        assert bool((la == lb)[clear].all()), name
        assert summary[f"argmax_agreement_{name}"] >= 0.99, summary
    rows = []
    for k, ref in g32.items():
        if k.endswith("conv.bias") and "sr_head" not in k or float(ref.norm()) == 0.0:
            continue
        cos = float((g16[k].double() * ref.double()).sum() / (g16[k].double().norm() * ref.double().norm()))
        rows.append((k, _l2rel(g16[k], ref), cos))
    rows.sort(key=lambda r: -r[1])
    summary["gradient_l2rel_worst"] = rows[:6]
    summary["gradient_l2rel_median"] = rows[len(rows) // 2][1]
    summary["gradient_cosine_min"] = min(r[2] for r in rows)
    print("[cfg5] " + json.dumps(summary))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_cfg5.json"), "w") as f:
        json.dump({"summary": summary, "gradients": rows}, f, indent=1)
    assert summary["loss_rel"] <= 1e-3 and summary["teacher_features_l2rel"] <= 1e-2
    assert summary["logits_l2rel_lr"] <= 3e-2 and summary["logits_l2rel_hr"] <= 3e-2
    for k, e, cos in rows:
        shallow = any(t in k for t in ("sr_head", "seg_layers", "distiller", "decoder.stages.4"))
        assert e <= (2e-2 if shallow else 0.6) and cos >= 0.8, (k, e, cos)
