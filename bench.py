#!/usr/bin/env python3
"""Headline benchmark: 3D patches/sec, FLAVR UNet_3D_3D training step on one
1x1x128x128x128 fp32 patch per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  N > 1, started plainly: this process never touches a GPU; it starts N ranks itself
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>),
  lets rank 0's JSON line through and exits with the launcher's code.  Started BY torch.distributed.run
  (RANK / WORLD_SIZE in the environment) it is one of those ranks.

A step = x.clone -> forward -> L1 loss -> backward -> (N>1: RCCL gradient all-reduce)
-> Adam update.  Inputs are synthetic and resident in HBM before the timed region.
Prints ONE JSON line on rank 0.

Order inside a rank: build -> W warm-up steps -> barrier + synchronize -> K timed steps (nothing but the step and one
HIP event per step boundary on the launch stream) -> synchronize + barrier -> [rank 0, N=1: CPU oracle leg] ->
a separate, untimed pass with HIP events around every matrix-core launch (the roofline families).  `value` is
patches / wall-clock of the K timed steps (max over ranks); `step_ms` gives median / min / max of the K steps.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16)


def build_model(n_inputs, dev):
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    torch.manual_seed(0)
    return UNet_3D_3D(img_channels=1, block="unet_18", n_inputs=n_inputs, n_outputs=4).to(dev)


def build_seg_model(dev):
    """BASELINE.json configs[2]: nnU-Net 3d_fullres SegModel, isotropic plan, upscale 4."""
    import torch.nn as nn
    from rehrseg_amd.models.seg_model import SegModel
    torch.manual_seed(0)
    return SegModel(input_channels=1, num_classes=2, n_stages=6, upscale=4,
                    features_per_stage=[32, 64, 128, 256, 320, 320], conv_op=nn.Conv3d,
                    kernel_sizes=[[3, 3, 3]] * 6, strides=[[1, 1, 1]] + [[2, 2, 2]] * 5, n_conv_per_stage=[2] * 6,
                    n_conv_per_stage_decoder=[2] * 5, conv_bias=True, norm_op=nn.InstanceNorm3d,
                    norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None, dropout_op_kwargs=None,
                    nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True}, deep_supervision=False).to(dev)


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); logical count if unknown."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def cpu_baseline(size, state_dict, x, tgt, timed=3):
    """The oracle (a CPU port of the reference's path, oracle/flavr_oracle.py) on the host cores: the SAME weights,
    input and target as the GPU model, 1 warm-up + `timed` forward+backward steps, median.  Returns the record and
    the warm-up step's (output, loss) for the parity line."""
    from oracle import flavr_oracle as fo
    threads = torch.get_num_threads()
    times, first = [], None
    for it in range(1 + timed):
        sd = {k: v.detach().clone().requires_grad_() for k, v in state_dict.items()}
        t0 = time.perf_counter()
        out = fo.unet_3d_3d(sd, x.clone(), 1, size, 4)
        loss = (out - tgt).abs().mean()
        loss.backward()
        dt = time.perf_counter() - t0
        if it == 0:
            first = (out.detach(), float(loss.detach()))
        else:
            times.append(dt)
        del sd, out, loss
    times.sort()
    med = times[len(times) // 2]
    rec = {"value": 1.0 / med, "unit": "patches/s", "cores": threads, "physical_cores": physical_cores(),
           "logical_cpus": os.cpu_count(), "kind": "port",
           "sample": f"forward+backward of the same 1x1x{size}^3 fp32 workload (same weights, input and target as the GPU "
                     f"model) through oracle/flavr_oracle.py on {threads} torch threads: 1 warm-up + {timed} timed steps, "
                     f"median {med:.1f} s (min {times[0]:.1f}, max {times[-1]:.1f}); no optimizer step"}
    return rec, first


def self_launch(n):
    """`python bench.py --gpus N` started plainly: start the N ranks as ONE child job of torch.distributed.run with
    the flags this process got, from a parent that has made no GPU call (importing torch makes none).  Rank 0's JSON
    line goes to the inherited stdout; returns the launcher's exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128, help="patch edge (128 = BASELINE config)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default=None,
                    help="bf16 = ops.mixed_precision(): bf16 operands on the matrix cores, fp32 accumulate / statistics / "
                         "master weights (BASELINE.json configs[4]); default fp32 (bf16 for --workload cfg5)")
    ap.add_argument("--workload", choices=["flavr", "seg", "flavr_ref", "cfg4", "cfg5", "stub"], default="flavr",
                    help="flavr = configs[1] (headline); seg = configs[2] (SegModel 2x1x128^3, secondary); "
                         "flavr_ref = the reference's own stage-1 training shape, UNet_3D_3D(2,..,4,4) on "
                         "(B,2,4,96,96) with the UASR head (configs/brain.yaml)")
    ap.add_argument("--batch", type=int, default=32, help="batch of the flavr_ref workload (brain.yaml: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps of the cpu_baseline leg (after 1 warm-up)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-wgrad-stream", action="store_true",
                    help="weight gradients on the main stream (default: PatchParallel launches them on a second HIP stream); "
                         "the per-kernel pass always runs without it, and the rocprofv3 kernel statistics under profiles/ "
                         "are collected with this flag so that kernel durations are not inflated by a co-running kernel")
    ap.add_argument("--no-teacher-stream", action="store_true",
                    help="cfg4 / cfg5: the frozen teacher's pass on the main stream (default: a second HIP stream, next to "
                         "the student's forward)")
    ap.add_argument("--no-form-cache", action="store_true",
                    help="frozen networks (cfg4 / cfg5: the teacher) re-pack their weights at every launch too")
    ap.add_argument("--kernel-steps", type=int, default=0,
                    help="steps of the separate per-kernel timing pass (0: max(--steps, 50), at most 100)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --workload stub: the launch / exchange / report plumbing on CPU ranks (tests)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # one rank of a torch.distributed.run job
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args.gpus))                            # before anything touches a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    on_cpu = args.workload == "stub" and args.backend == "gloo"
    if args.workload == "stub" and not on_cpu:
        raise SystemExit("--workload stub is the CPU plumbing test: use it with --backend gloo")
    # --backend gloo with a real workload = REHEARSAL of the N > 1 path on a box with fewer GPUs than ranks: the ranks share
    # the devices round-robin and gloo carries the gradient buckets through the host.  The line it prints is not a
    # measurement (config.rehearsal says so); the product path itself still has no CPU fallback.
    rehearsal = args.backend == "gloo" and not on_cpu
    n_dev = 1 if on_cpu else max(1, torch.cuda.device_count())   # (counting devices initialises nothing)
    dev_index = local_rank % n_dev if rehearsal else local_rank
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not on_cpu:
            torch.cuda.set_device(dev_index)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if on_cpu:
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", dev_index)
        torch.cuda.set_device(dev)

    from rehrseg_amd import hip_backend
    from rehrseg_amd.parallel import PatchParallel
    hip_backend.WEIGHT_FORM_CACHE = not args.no_form_cache

    if args.workload == "cfg5":   # BASELINE.json configs[4]: the joint step in bf16 mixed precision at 160^3
        args.workload = "cfg4"
        args.precision = args.precision or "bf16"
        if args.size == 128:
            args.size = 160
        cfg5 = True
    else:
        cfg5 = False
    mixed = args.precision == "bf16"
    from rehrseg_amd import ops
    size = args.size
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)  # every rank draws its own patch
    streams = {"teacher": not args.no_teacher_stream}     # (the per-kernel pass switches the second streams off)
    if args.workload == "stub":
        # plumbing only (tests): a two-layer torch Conv3d net, no HIP kernels; exercises the launch, the flat-bucket
        # exchange and the report with CPU ranks
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ReLU(),
                                    torch.nn.Conv3d(4, 1, 3, padding=1)).to(dev)
        pp = PatchParallel(model, bucket_mb=1e-4, direct=[], wgrad_stream=False)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        x = torch.rand(1, 1, 8, 8, 8, generator=g).to(dev)
        tgt = torch.rand(1, 1, 8, 8, 8, generator=g).to(dev)
        patches_per_step = 1
        workload = "stub: 2-layer torch Conv3d net on 1x1x8^3 (launch / exchange / report plumbing, not a measurement)"

        def step():
            pp.zero_grad()
            loss = (model(x) - tgt).abs().mean()
            loss.backward()
            pp.reduce_gradients()
            opt.step()
            return loss
    elif args.workload == "flavr":
        model = build_model(size, dev)
        pp = PatchParallel(model, wgrad_stream=not args.no_wgrad_stream)
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
        x = torch.rand(1, 1, size, size, size, generator=g).to(dev)
        tgt = torch.rand(1, 1, 4, size, size, generator=g).to(dev)
        patches_per_step = 1
        workload = (f"FLAVR UNet_3D_3D(1,'unet_18',{size},4) fwd+bwd+Adam, 1x1x{size}^3 patch per GPU, "
                    "random-init weights")

        def step():
            pp.zero_grad()
            out = model(x.clone())  # forward subtracts the mean in place, like the reference
            loss = (out - tgt).abs().mean()
            loss.backward()
            pp.reduce_gradients()
            opt.step()
            return loss
    elif args.workload == "flavr_ref":
        from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
        from rehrseg_amd.train_steps import train_sr_step
        from rehrseg_amd.utils.seg_utils import BCEDiceLoss
        torch.manual_seed(0)
        model = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(dev)
        pp = PatchParallel(model, wgrad_stream=not args.no_wgrad_stream)
        opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.99), fused=True)
        B = args.batch
        x = torch.rand(B, 2, 4, 96, 96, generator=g).to(dev)
        hr = torch.rand(B, 2, 16, 96, 96, generator=g)
        hr[:, 1:] = (hr[:, 1:] > 0.5).float()
        hr = hr.to(dev)
        patches_per_step = B
        workload = (f"FLAVR UNet_3D_3D(2,'unet_18',4,4,use_uncertainty) train_sr step (L1 + UASR terms + BCEDice + Adam), "
                    f"{B}x2x4x96x96 per GPU, random-init weights")
        l1, bd = torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0)

        def step():
            return train_sr_step(model, opt, None, x.clone(), hr, l1, bd, 4.0, 4, True,
                                 grad_sync=pp.reduce_gradients)
    elif args.workload == "cfg4":
        # BASELINE.json configs[3]: the joint stage-2 step (train_all.py:519-556), 1 LR patch per GPU: frozen FLAVR
        # teacher (D-1 windows, batched, truncated after the consumed level) + SegModel student on the
        # distillation-compatible anisotropic plan + Distiller, uncertainty-weighted CE / DC+CE, SGD
        import itertools
        import torch.nn as nn
        from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
        from rehrseg_amd.models.seg_model import Distiller, SegModel
        from rehrseg_amd.train_steps import train_segsr_step
        from rehrseg_amd.utils.seg_utils import _build_loss
        torch.manual_seed(0)
        teacher = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(dev).eval()
        for q in teacher.parameters():
            q.requires_grad_(False)
        student = SegModel(input_channels=1, num_classes=2, n_stages=6, upscale=4,
                           features_per_stage=[32, 64, 128, 256, 320, 320], conv_op=nn.Conv3d,
                           kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
                           strides=[[1, 1, 1], [1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]],
                           n_conv_per_stage=[2] * 6, n_conv_per_stage_decoder=[2] * 5, conv_bias=True,
                           norm_op=nn.InstanceNorm3d, norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None,
                           dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True},
                           deep_supervision=False).to(dev)
        dist_m = Distiller(64, 64, lambda_l1=0.0, lambda_cosine=1.0, lambda_structure=1.0).to(dev)
        model = nn.ModuleDict({"student": student, "distiller": dist_m})
        pp = PatchParallel(model, wgrad_stream=not args.no_wgrad_stream)
        opt = torch.optim.SGD(itertools.chain(student.parameters(), dist_m.parameters()), lr=1e-2, momentum=0.99,
                              nesterov=True, weight_decay=3e-5)
        img = torch.randn(1, 1, size, size, size, generator=g).to(dev)
        lab_lr = torch.randint(0, 2, (1, 1, size, size, size), generator=g).float().to(dev)
        lab_hr = torch.randint(0, 2, (1, 1, 4 * size, size, size), generator=g).float().to(dev)
        unc = (1.0 - torch.randint(0, 256, (1, 1, size, size, size), generator=g).float() / 255.0 * 0.99).to(dev)
        patches_per_step = 1
        workload = (f"joint stage-2 step: FLAVR teacher (no-grad, {size - 1} windows batched) + SegModel student "
                    f"(anisotropic plan, upscale 4) + Distiller + SGD, 1x1x{size}^3 LR patch per GPU, random-init weights")
        l_lr, l_hr = _build_loss(False, weight_dice=0), _build_loss(False, weight_dice=1)

        def step():
            return train_segsr_step(student, teacher, dist_m, opt, img.clone(), lab_lr, lab_hr, unc, l_lr, l_hr,
                                    grad_sync=pp.reduce_gradients, teacher_stream=streams["teacher"])
    else:
        model = build_seg_model(dev)
        pp = PatchParallel(model, wgrad_stream=not args.no_wgrad_stream)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
        x = torch.randn(2, 1, size, size, size, generator=g).to(dev)
        lab_lr = torch.randint(0, 2, (2, 1, size, size, size), generator=g).float().to(dev)
        lab_hr = torch.randint(0, 2, (2, 1, 4 * size, size, size), generator=g).float().to(dev)
        patches_per_step = 2
        workload = (f"SegModel (nnU-Net 3d_fullres isotropic plan, upscale 4) fwd+bwd+SGD, 2x1x{size}^3 per GPU, "
                    "DC+CE (fused HIP loss) on LR and HR logits, random-init weights")
        from rehrseg_amd.utils.seg_utils import _build_loss
        ce = _build_loss()   # BASELINE cfg-3: Dice + CE on both heads (utils/seg_utils.py:353-372)

        def step():
            pp.zero_grad()
            out, out_up = model(x)
            loss = ce(out, lab_lr) + ce(out_up, lab_hr)
            loss.backward()
            pp.reduce_gradients()
            opt.step()
            return loss

    if mixed:
        plain_step = step

        def step():
            with ops.mixed_precision():
                return plain_step()
        workload = "[bf16 mixed precision: bf16 MFMA operands, fp32 accumulate / statistics / master weights] " + workload
    def sync():
        if not on_cpu:
            torch.cuda.synchronize()

    # the CPU leg needs the model's INITIAL weights (host copy now; everything else of that leg -- the HIP forward for
    # the parity line included -- runs after the timed region: a no-grad forward here leaves the caching allocator
    # with blocks that one later training step has to re-malloc, a 40 ms outlier inside the timed region)
    cpu_leg = world == 1 and not args.no_cpu_baseline and args.workload == "flavr" and not mixed
    if cpu_leg:
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for _ in range(args.warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    sync()
    # ---- the timed region: exactly K steps; per-step boundaries are HIP events on the launch stream
    marks = None if on_cpu else [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # Python's cyclic collector: a full (generation-2) pass walks every live object of the process and stalls the
    # launching thread for tens of ms -- one 40 ms step per ~100.  The long-lived objects (model, optimizer, torch) are
    # moved out of the collector's sight once, before the clock starts; young generations keep being collected.
    gc.collect()
    gc.freeze()
    gc_log = []
    gc_cb = lambda phase, info: gc_log.append((info["generation"], time.perf_counter())) if phase == "stop" else None
    gc.callbacks.append(gc_cb)
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    for i in range(args.steps):
        loss = step()
        if marks:
            marks[i + 1].record()
    sync()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.callbacks.remove(gc_cb)
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    raw_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)] if marks else []
    step_ms = sorted(raw_ms)
    final_loss = float(loss.item())

    # ---- CPU leg (rank 0, N=1, headline workload): the oracle on the host cores with the GPU model's own initial
    # weights and patch; its warm-up step doubles as the parity check of the HIP forward (north_star: within 1e-3)
    cpu_rec = parity = None
    if cpu_leg:
        cpu_rec, (ref_out, ref_loss) = cpu_baseline(size, sd0, x.cpu(), tgt.cpu(), args.cpu_steps)
        trained = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model.load_state_dict(sd0)                               # the weights the oracle just ran with
        with torch.no_grad():
            out0 = model(x.clone())
            loss0 = float((out0 - tgt).abs().mean())
        out0 = out0.cpu()
        model.load_state_dict(trained)
        del trained
        parity = {"fwd_rel": float((out0 - ref_out).abs().max() / ref_out.abs().max()),
                  "loss_rel": abs(loss0 - ref_loss) / abs(ref_loss), "loss_hip": loss0, "loss_cpu": ref_loss,
                  "what": "HIP forward vs the CPU oracle's forward on identical (initial) weights / input"}
        del sd0, ref_out, out0

    # ---- per-kernel pass, untimed: HIP events around every matrix-core launch.  Separate from the timed region
    # (the events cost host time per launch) and LAST in the process, after the CPU leg.
    prof, ksteps = {}, 0
    if not args.no_kernel_timing and not on_cpu:
        ksteps = args.kernel_steps or min(100, max(args.steps, 50))
        pp.set_wgrad_stream(False)                  # one kernel at a time on the chip: un-shared durations
        streams["teacher"] = False
        for _ in range(2):
            step()                                  # the GPU idled through the CPU leg
        torch.cuda.synchronize()
        hip_backend.profile_start()
        for _ in range(ksteps):
            step()
        prof = hip_backend.profile_stop()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rec = {
            "metric": f"3D patches/sec (fwd+bwd, {size}^3 {'bf16 mixed precision' if mixed else 'fp32'})", "value": world * patches_per_step * args.steps / elapsed,
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if mixed else "f32",
            "data": "synthetic",
            "config": {"workload": workload, "patches_per_gpu": patches_per_step,
                       "global_batch": world * patches_per_step,
                       "parallelism": f"dp{world} (patch-parallel, flat-bucket RCCL all-reduce)"
                                      + (" -- REHEARSAL: gloo ranks sharing devices, not a measurement" if rehearsal else "")},
            "loss": final_loss,
        }
        if step_ms:
            rec["step_ms"] = {"median": step_ms[len(step_ms) // 2], "min": step_ms[0], "max": step_ms[-1],
                              "slowest_step_index": raw_ms.index(step_ms[-1]),
                              "gc_passes_in_timed_region": {str(g_): sum(1 for q in gc_log if q[0] == g_) for g_ in (0, 1, 2)},
                              "what": "HIP-event time of each of the K timed steps on the launch stream"}
            rec["value_at_median_step"] = world * patches_per_step / (rec["step_ms"]["median"] * 1e-3)
        EXEC = {"wino_conv": 16.0 / 36.0, "wino_wgrad": 16.0 / 36.0,      # F(2x2,3x3): 16 of 36 products issued
                "wino22_conv": 9.0 / 16.0, "wino22_wgrad": 9.0 / 16.0,    # F(2x2,2x2): 9 of 16
                "gather_gemm": 1.0, "wgrad": 1.0, "gather_gemm_bf16": 1.0, "wgrad_bf16": 1.0}
        PEAK = {"gather_gemm_bf16": BF16_MFMA_PEAK_TFLOPS, "wgrad_bf16": BF16_MFMA_PEAK_TFLOPS}

        def fam(name):
            """Roofline record of a kernel family.  `achieved` / `frac` price the MFMA work the kernels ISSUE
            (a fraction <= 1 of the matrix pipe); `algorithmic_*` price the direct-convolution FLOPs the layer
            needs, which the Winograd kernels reach with 16/36 resp. 9/16 of the products."""
            f = prof.get(name)
            if not f or f["seconds"] <= 0:
                return None
            alg = f["flops"] / f["seconds"] / 1e12
            ex = alg * EXEC[name]
            peak = PEAK.get(name, FP32_MFMA_PEAK_TFLOPS)
            return {"bound": "mfma", "achieved": ex, "peak": peak, "unit": "TFLOP/s",
                    "frac": ex / peak, "traffic": None,
                    "algorithmic_achieved": alg, "algorithmic_frac": alg / peak,
                    "mfma_products_issued_per_algorithmic": EXEC[name],
                    "launches_per_step": f["launches"] / ksteps,
                    "avg_launch_ms": f["seconds"] / f["launches"] * 1e3,
                    "ms_per_step": f["seconds"] / ksteps * 1e3,
                    "algorithmic_gflop_per_step": f["flops"] / ksteps / 1e9}

        names = {"wino_conv": "wino_conv_big8_kernel / wino_conv_w32p_kernel / wino_flat8_conv_kernel / wino_conv_kernel: "
                              "Winograd F(2x2,3x3)-over-(H,W) fp32 MFMA conv forward / input gradient",
                 "gather_gemm": "gather_gemm_kernel / halo_conv_kernel: fp32 MFMA implicit-GEMM conv (strided, 1x1x1, "
                                "transposed phases)",
                 "gather_gemm_bf16": "gather_gemm_bf16_kernel: bf16-operand / fp32-accumulate implicit-GEMM conv forward, "
                                     "input gradient, transposed-conv phases (v_mfma_f32_32x32x16_bf16)",
                 "wgrad_bf16": "wgrad_bf16_kernel: bf16-operand / fp32-accumulate weight gradient"}
        fams = {n: fam(n) for n in EXEC}
        dominant = max((n for n in fams if fams[n]), key=lambda n: fams[n]["ms_per_step"], default=None)
        if dominant:
            r = fams[dominant]
            r["kernel"] = names.get(dominant, dominant)
            # HBM bytes per launch from the committed PMC passes of the same command (rocprofv3 counters cannot be
            # collected from inside the timed run): tools/pmc_hbm.py over three rocprofv3 passes of
            # `bench.py --workload flavr` (tools/collect_profiles_r03.sh pmc), newest round first
            if args.workload == "flavr" and size == 128 and dominant == "wino_conv":
                for tag in ("r03", "r02"):
                    tp = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_flavr.json")
                    if not os.path.exists(tp):
                        continue
                    k = json.load(open(tp)).get("kernels", {}).get("wino_conv_big8_kernel") or {}
                    if k.get("hbm_bytes_per_launch"):
                        r["traffic"] = k["hbm_bytes_per_launch"]
                        r["traffic_source"] = (f"profiles/{tag}_pmc_hbm_flavr.json (PMC FETCH_SIZE x2 + WRITE_SIZE per launch "
                                               "of wino_conv_big8_kernel; separate rocprofv3 passes of this command)")
                        break
            rec["roofline"] = r
        for n, r in fams.items():
            if r and n != dominant:
                rec["roofline_" + n] = r
        if prof:
            rec["mfma_kernel_ms_per_step"] = sum(v["seconds"] for v in prof.values()) / ksteps * 1e3
            rec["kernel_timing"] = (f"HIP events around every matrix-core launch over {ksteps} further steps after the "
                                    "timed region (not inside it), weight gradients on the main stream: a kernel's "
                                    "duration while it shares the chip with the other stream's kernel is not its own")
        rec["config"]["wgrad_stream"] = not args.no_wgrad_stream
        if cpu_rec is not None:
            rec["cpu_baseline"] = cpu_rec
            rec["parity_vs_cpu"] = parity
        print(json.dumps(rec), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
