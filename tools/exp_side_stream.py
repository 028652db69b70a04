"""Experiment: weight gradients on a side HIP stream (they are independent of the input-gradient chain), flavr / seg
workloads of bench.py.  python tools/exp_side_stream.py flavr|seg"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench
from rehrseg_amd import ops
from rehrseg_amd.parallel import PatchParallel

which = sys.argv[1] if len(sys.argv) > 1 else "flavr"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
if which == "flavr":
    model = bench.build_model(128, dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
    x = torch.rand(1, 1, 128, 128, 128, generator=g).to(dev)
    tgt = torch.rand(1, 1, 4, 128, 128, generator=g).to(dev)
    loss_fn = lambda: (model(x.clone()) - tgt).abs().mean()
else:
    from rehrseg_amd.utils.seg_utils import _build_loss
    model = bench.build_seg_model(dev)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).to(dev)
    lab_lr = torch.randint(0, 2, (2, 1, 128, 128, 128), generator=g).float().to(dev)
    lab_hr = torch.randint(0, 2, (2, 1, 512, 128, 128), generator=g).float().to(dev)
    ce = _build_loss()

    def loss_fn():
        out, out_up = model(x)
        return ce(out, lab_lr) + ce(out_up, lab_hr)
pp = PatchParallel(model)
side = torch.cuda.Stream()
orig = ops.conv_wgrad
use_side = False


def wg(dz, x1, x2, w, cfg, want_bias, out=None):
    if not use_side:
        return orig(dz, x1, x2, w, cfg, want_bias, out=out)
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        r = orig(dz, x1, x2, w, cfg, want_bias, out=out)
    for t in (dz, x1, x2, out):
        if t is not None:
            t.record_stream(side)
    return r


ops.conv_wgrad = wg


def step():
    pp.zero_grad()
    loss = loss_fn()
    loss.backward()
    if use_side:
        torch.cuda.current_stream().wait_stream(side)
    pp.reduce_gradients()
    opt.step()
    return loss


for mode in (False, True, False, True):
    use_side = mode
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loss = step()
    torch.cuda.synchronize()
    print(which, "side-stream wgrad" if mode else "single stream   ", f"{(time.perf_counter() - t0) / 30 * 1e3:.2f} ms/step  loss {float(loss):.6f}", flush=True)
