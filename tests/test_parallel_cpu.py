"""PatchParallel (flat-bucket gradient all-reduce) with world_size 2 over gloo on the CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rehrseg_amd.parallel import PatchParallel
    torch.manual_seed(100 + rank)  # different init per rank: the wrapper must broadcast rank 0's
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    pp = PatchParallel(model, bucket_mb=3e-5)  # ~7 elements per bucket: several buckets, launched from hooks
    assert len(pp.buckets) >= 2 and pp.overlap
    w0 = torch.cat([p.detach().flatten() for p in model.parameters()])
    torch.manual_seed(7 + rank)  # every rank draws its own "patch"
    x = torch.randn(4, 6)
    outs = []
    for _ in range(2):
        pp.zero_grad()
        model(x).square().mean().backward()
        pp.reduce_gradients()
        outs.append(pp.flat.clone())
    q.put((rank, w0.numpy(), x.numpy(), [o.numpy() for o in outs]))  # plain arrays: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_patch_parallel_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(w), torch.from_numpy(x), [torch.from_numpy(o) for o in g]) for r, w, x, g in res]
    (_, w0a, xa, ga), (_, w0b, xb, gb) = res
    assert torch.equal(w0a, w0b)                      # parameters were broadcast from rank 0
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, atol=1e-7)         # both ranks hold the averaged gradient
    # reference: mean of the two per-rank gradients, computed in one process
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    gs = []
    for x in (xa, xb):
        model.zero_grad()
        model(x).square().mean().backward()
        gs.append(torch.cat([p.grad.flatten() for p in reversed(list(model.parameters()))]))  # flat buffer order
    assert torch.allclose(ga[0], (gs[0] + gs[1]) / 2, atol=1e-6)
    assert torch.allclose(ga[1], ga[0], atol=1e-7)     # zero_grad keeps the views; step 2 == step 1


def test_single_process_is_a_noop_reduce():
    from rehrseg_amd.parallel import PatchParallel
    m = torch.nn.Linear(3, 2)
    pp = PatchParallel(m)
    m(torch.ones(1, 3)).sum().backward()
    before = pp.flat.clone()
    pp.reduce_gradients()
    assert torch.equal(before, pp.flat)
    assert m.bias.grad.data_ptr() == pp.flat.data_ptr()  # reverse registration order: the last parameter comes first


# ----------------------------------------------------------------------------- every gradient route, world 2
class _Routes(torch.nn.Module):
    """conv_a: 5-D weight used once (direct in-place write by the weight-gradient kernel) + bias (stolen);
    conv_b: 5-D weight used TWICE (must keep autograd's accumulation); norm: stolen small parameters;
    unused: never part of the graph (contributes zeros)."""

    def __init__(self):
        super().__init__()
        self.conv_a = torch.nn.Conv3d(16, 16, 3, padding=1)
        self.conv_b = torch.nn.Conv3d(16, 16, 3, padding=1, bias=False)
        self.norm = torch.nn.InstanceNorm3d(16, affine=True)
        self.unused = torch.nn.Linear(3, 3)

    def forward(self, x, conv):
        y = conv(x, self.conv_a.weight, self.conv_a.bias)
        y = conv(y, self.conv_b.weight, None)
        y = conv(y, self.conv_b.weight, None)
        return torch.nn.functional.instance_norm(y, weight=self.norm.weight, bias=self.norm.bias)


def _routes_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_backend
    from rehrseg_amd import ops
    from rehrseg_amd.parallel import PatchParallel
    ops.set_backend(emu_backend)
    torch.manual_seed(3)
    m = _Routes()
    # small buckets: several exchanges launched from hooks; the conv weights take the direct route
    pp = PatchParallel(m, bucket_mb=2e-5, direct=[m.conv_a.weight, m.conv_b.weight])
    assert pp.overlap and len(pp.buckets) >= 3
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    torch.manual_seed(50 + rank)
    x = torch.randn(1, 16, 3, 6, 6)
    g = torch.randn(1, 16, 3, 6, 6)
    conv = lambda t, w, b: ops.fused_conv3d(t, w, b, 1, 1)
    grads, wrote = [], []
    for step in range(3):
        if step == 1:
            opt.zero_grad()            # a plain optimizer.zero_grad(set_to_none=True): drops every .grad view
        else:
            pp.zero_grad()
        (m(x, conv) * g).mean().backward()
        wrote.append((pp.was_written(m.conv_a.weight), pp.was_written(m.conv_b.weight)))
        pp.reduce_gradients()
        grads.append({n: p.grad.detach().clone().numpy() for n, p in m.named_parameters()})
        opt.step()
    q.put((rank, x.numpy(), g.numpy(), grads, wrote, {n: p.detach().numpy() for n, p in m.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_every_gradient_route_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_routes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, xa, ga, grads_a, wrote_a, pa), (_, xb, gb, grads_b, wrote_b, pb) = res
    # step 0 and 2 (pp.zero_grad): conv_a's weight is written in place by the kernel, the twice-used conv_b is not;
    # step 1 (plain optimizer.zero_grad): no views to write into, everything goes through autograd + reconcile
    assert wrote_a == [(True, False), (False, False), (True, False)] == wrote_b
    for n in pa:
        assert (pa[n] == pb[n]).all(), n                                  # the ranks stay in lockstep
    # reference: plain torch, the mean of the two per-rank gradients, three SGD steps
    torch.manual_seed(3)
    ref = _Routes()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    conv = lambda t, w, b: torch.nn.functional.conv3d(t, w, b, 1, 1)
    for step in range(3):
        per_rank = []
        for x, g in ((xa, ga), (xb, gb)):
            ref.zero_grad()
            (ref(torch.from_numpy(x), conv) * torch.from_numpy(g)).mean().backward()
            per_rank.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                             for n, p in ref.named_parameters()})
        for n, p in ref.named_parameters():
            want = (per_rank[0][n] + per_rank[1][n]) / 2
            got = torch.from_numpy(grads_a[step][n])
            assert torch.allclose(got, want, atol=1e-6, rtol=1e-4), (step, n)
            assert (grads_a[step][n] == grads_b[step][n]).all()
            p.grad = want
        opt.step()


def _sr_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(3)   # two ranks share the CPU suite's cores
    import emu_backend
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd import ops
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    from rehrseg_amd.parallel import PatchParallel
    from rehrseg_amd.train_steps import train_sr_step
    from rehrseg_amd.utils.seg_utils import BCEDiceLoss
    ops.set_backend(emu_backend)
    m = UNet_3D_3D(2, "unet_18", 4, 4)
    m.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()})
    pp = PatchParallel(m, bucket_mb=16, direct=[p for p in m.parameters() if p.dim() == 5])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)            # a RAW optimizer: the step function must cope
    lr_p = det_input(f"pp.lr{rank}", (1, 2, 4, 16, 16), "rand")
    hr_p = det_input(f"pp.hr{rank}", (1, 2, 16, 16, 16), "rand")
    hr_p[:, 1:] = (hr_p[:, 1:] > 0.5).float()
    losses = []
    for _ in range(2):
        losses.append(float(train_sr_step(m, opt, None, lr_p.clone(), hr_p, torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0),
                                          4.0, 4, False, grad_sync=pp.reduce_gradients)))
    q.put((rank, losses, len(pp.buckets),
           {n: p.detach().numpy() for n, p in m.named_parameters() if n in
            ("encoder.stem.0.weight", "decoder.4.upconv.0.weight", "outconv.1.bias", "encoder.layer3.0.conv1.0.weight")}))
    dist.barrier()
    dist.destroy_process_group()


def test_train_sr_step_with_raw_optimizer_keeps_ranks_in_lockstep():
    """train_sr_step + PatchParallel.reduce_gradients + a plain optimizer must not let the ranks
    diverge (zero_grad(set_to_none=True) used to drop the flat views)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sr_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, la, nb, pa), (_, lb, _, pb) = res
    assert nb >= 2
    assert la[0] != lb[0]                                     # different patches per rank ...
    for n in pa:
        assert (pa[n] == pb[n]).all(), n                      # ... identical parameters after two steps
    assert la[1] != la[0]


def test_direct_route_survives_zero_grad_after_forward():
    """the reference's loops clear gradients AFTER the forward (train_all.py:135-137); the use counts
    of that forward must survive PatchParallel.zero_grad(), or the weight-gradient kernels never write in place.
    Single process, ABI emulation; gradients equal those of the run without PatchParallel."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import emu_backend
    from oracle.detinit import det_input, det_tensor
    from rehrseg_amd import ops
    from rehrseg_amd.models.FLAVR.FLAVR_arch import UNet_3D_3D
    from rehrseg_amd.parallel import PatchParallel
    from rehrseg_amd.train_steps import train_sr_step
    from rehrseg_amd.utils.seg_utils import BCEDiceLoss
    prev = ops.get_backend() if hasattr(ops, "get_backend") else None
    ops.set_backend(emu_backend)
    try:
        lr_p = det_input("pp.lr", (1, 2, 4, 16, 16), "rand")
        hr_p = det_input("pp.hr", (1, 2, 16, 16, 16), "rand")
        hr_p[:, 1:] = (hr_p[:, 1:] > 0.5).float()

        class _NoStep:                                           # keeps the parameters, so both runs see the same weights
            def zero_grad(self, set_to_none=True):
                for p in self.ps:
                    p.grad = None

            def step(self):
                pass

        def run(with_pp):
            m = UNet_3D_3D(2, "unet_18", 4, 4)
            m.load_state_dict({k: det_tensor(k, tuple(v.shape)) for k, v in m.state_dict().items()})
            opt = _NoStep()
            opt.ps = list(m.parameters())
            seen = {}
            if with_pp:
                direct = [p for p in m.parameters() if p.dim() == 5]
                pp = PatchParallel(m, direct=direct)

                def sync():
                    seen["written"] = sum(pp.was_written(p) for p in direct)
                    seen["direct"] = sum(1 for n, p in m.named_parameters()
                                         if p.dim() == 5 and "attn_layer" not in n)
                    pp.reduce_gradients()
                for _ in range(2):                               # the second step must behave like the first
                    train_sr_step(m, opt, None, lr_p.clone(), hr_p, torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0), 4.0, 4,
                                  False, grad_sync=sync, zero_grad=pp.zero_grad)
                pp.close()
            else:
                train_sr_step(m, opt, None, lr_p.clone(), hr_p, torch.nn.L1Loss(), BCEDiceLoss(1.0, 1.0), 4.0, 4, False)
            return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, seen

        g_pp, seen = run(True)
        g_ref, _ = run(False)
        # every conv weight that goes through conv_wgrad takes the in-place route (the thin-input stem through a copy
        # into its slot); the SEGating 1x1x1 weights get their gradient from se_gate_bwd
        assert seen["written"] == seen["direct"] == 25, seen
        assert set(g_pp) == set(g_ref) and len(g_ref) > 60
        for n in g_ref:
            assert torch.allclose(g_pp[n], g_ref[n], atol=1e-6, rtol=1e-5), n
    finally:
        if prev is not None:
            ops.set_backend(prev)
