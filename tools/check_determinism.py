"""Run-to-run reproducibility of the training path on the device: N times from the same weights, three SGD steps and
two accumulated backward passes (the sequence of tests/test_parallel_gpu.py::test_side_stream_...), every final
gradient / parameter compared with the first run's (max difference relative to the tensor's largest element).
usage: check_determinism.py [runs] [--side]"""
import sys
sys.path.insert(0, ".")
import torch
from oracle.detinit import det_tensor
from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
from rehrseg_amd.parallel import PatchParallel

dev = torch.device("cuda:0")
side = "--side" in sys.argv
runs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
x = torch.rand(2, 2, 4, 48, 40, generator=torch.Generator().manual_seed(5)).to(dev)
ref = None
for run in range(runs):
    m = UNet_3D_3D(2, "unet_18", 4, 4)
    m.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()})
    m = m.to(dev)
    pp = PatchParallel(m, wgrad_stream=side)
    opt = torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9)
    for _ in range(3):
        pp.zero_grad()
        m(x.clone()).abs().mean().backward()
        pp.reduce_gradients()
        opt.step()
    pp.zero_grad()
    for _ in range(2):
        m(x.clone()).abs().mean().backward()
    pp.reduce_gradients()
    torch.cuda.synchronize()
    g = {"grad " + n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    g.update({"param " + n: p.detach().clone() for n, p in m.named_parameters()})
    pp.close()
    if ref is None:
        ref = g
        continue
    bad = []
    for n in g:
        d = (g[n] - ref[n]).abs()
        bad.append((float(d.max()) / (float(ref[n].abs().max()) + 1e-30), n, int((d > 1e-5 * ref[n].abs().max()).sum()),
                    g[n].numel(), float(ref[n].abs().max())))
    bad.sort(reverse=True)
    print(f"run {run}: worst tensors vs run 0")
    for b in bad[:3]:
        print("   rel %.3e  %s  %d / %d elements off by > 1e-5, |t|max %.3e" % b)
