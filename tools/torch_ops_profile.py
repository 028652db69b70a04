"""Which torch (non-rehr) ops does a training step still launch, and on what shapes?  (GPU box)
    python tools/torch_ops_profile.py flavr_ref|flavr
Prints the aten ops of one step grouped by (name, input shapes), sorted by device time, with the python frame that
issued the largest of them."""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from torch.profiler import profile, ProfilerActivity  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "flavr_ref"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D  # noqa: E402
from rehrseg_amd.train_steps import train_sr_step  # noqa: E402
from rehrseg_amd.utils.seg_utils import BCEDiceLoss  # noqa: E402
torch.manual_seed(0)
if which == "flavr_ref":
    model = UNet_3D_3D(2, "unet_18", 4, 4, use_uncertainty=True).to(dev)
    x = torch.rand(32, 2, 4, 96, 96, generator=g).to(dev)
    hr = torch.rand(32, 2, 16, 96, 96, generator=g)
    unc = True
else:
    model = UNet_3D_3D(2, "unet_18", 4, 4).to(dev)
    x = torch.rand(1, 2, 4, 256, 256, generator=g).to(dev)
    hr = torch.rand(1, 2, 16, 256, 256, generator=g)
    unc = False
hr[:, 1:] = (hr[:, 1:] > 0.5).float()
hr = hr.to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.99), fused=True)
l1, bd = torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0)


def step():
    return train_sr_step(model, opt, None, x.clone(), hr, l1, bd, 4.0, 4, unc)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if not e.key.startswith("aten::"):
        continue
    t = getattr(e, "self_device_time_total", 0)
    if t <= 0:
        continue
    rows.append((t, e.count, e.key, str(e.input_shapes)[:110], [s for s in e.stack if "rehrseg_amd" in s or "tools" in s][:2]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("aten device time per step: %.2f ms" % (tot / 1e3))
for t, n, k, sh, st in rows[:45]:
    print("%8.1f us  x%-3d %-28s %s\n             %s" % (t, n, k, sh, " | ".join(s.split("/")[-1][:70] for s in st)))
