"""Per-launch times of the matrix-core kernels in one training step, by layer shape (GPU box):
    python tools/layer_times.py flavr_ref|flavr|seg
Uses hip_backend's event timing (the same events bench.py's roofline line sums per family)."""
import collections, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rehrseg_amd import hip_backend as hb  # noqa: E402
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D  # noqa: E402
from rehrseg_amd.train_steps import train_sr_step  # noqa: E402
from rehrseg_amd.utils.seg_utils import BCEDiceLoss  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "flavr_ref"
mixed = which.endswith("_bf16")          # e.g. seg_bf16: the same workload under ops.mixed_precision()
which = which[:-5] if mixed else which
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
torch.manual_seed(0)
if which == "seg":   # bench.py's cfg-3 workload
    sys.path.insert(0, root)
    import bench
    from rehrseg_amd.utils.seg_utils import _build_loss
    model = bench.build_seg_model(dev)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).to(dev)
    lab_lr = torch.randint(0, 2, (2, 1, 128, 128, 128), generator=g).float().to(dev)
    lab_hr = torch.randint(0, 2, (2, 1, 512, 128, 128), generator=g).float().to(dev)
    ce = _build_loss()
elif which == "flavr_ref":
    model, unc = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(dev), True
    x, hr = torch.rand(32, 2, 4, 96, 96, generator=g).to(dev), torch.rand(32, 2, 16, 96, 96, generator=g)
else:   # bench.py's headline workload (BASELINE cfg-2)
    model = UNet_3D_3D(img_channels=1, block="unet_18", n_inputs=128, n_outputs=4).to(dev)
    x, tgt = torch.rand(1, 1, 128, 128, 128, generator=g).to(dev), torch.rand(1, 1, 4, 128, 128, generator=g).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.99), fused=True)
if which == "seg":
    def step():
        opt.zero_grad(set_to_none=True)
        out, out_up = model(x)
        (ce(out, lab_lr) + ce(out_up, lab_hr)).backward()
        opt.step()
elif which == "flavr_ref":
    hr[:, 1:] = (hr[:, 1:] > 0.5).float()
    hr = hr.to(dev)
    l1, bd = torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0)
    step = lambda: train_sr_step(model, opt, None, x.clone(), hr, l1, bd, 4.0, 4, unc)  # noqa: E731
else:
    def step():
        opt.zero_grad(set_to_none=True)
        (model(x.clone()) - tgt).abs().mean().backward()
        opt.step()
if mixed:
    from rehrseg_amd import ops  # noqa: E402
    plain_step = step

    def step():
        with ops.mixed_precision():
            plain_step()
for _ in range(3):
    step()
acc = collections.OrderedDict()
REP = 3
for _ in range(REP):
    hb.profile_start()
    step()
    for fam, tag, flops, sec in hb.profile_stop(layers=True):
        d = acc.setdefault((fam, tag), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += sec
rows = sorted(acc.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for _, v in rows) / REP
print(f"matrix-core launches: {tot * 1e3:.2f} ms/step")
for (fam, tag), (n, fl, sec) in rows[:60]:
    print(f"{sec / REP * 1e3:7.3f} ms  x{n // REP:<2d} {fl / sec / 1e12:6.1f} TF alg  {fam:13s} {tag}")
