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
