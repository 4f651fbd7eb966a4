"""Data-parallel plumbing on CPU: two gloo ranks (no GPU needed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters, shard_range
        torch.manual_seed(100 + rank)  # replicas start different on purpose
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 3, padding=1))
        broadcast_parameters(net, src=0)
        w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        red = FlatGradAllReducer(net.parameters())
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
        # global batch of 4 samples, sharded 2 per rank
        g = torch.Generator().manual_seed(7)
        x_all, y_all = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 2, 8, 8, generator=g)
        idx = list(shard_range(4, rank, world))
        for _ in range(2):
            red.zero_grad()
            loss = torch.nn.functional.mse_loss(net(x_all[idx]), y_all[idx])
            loss.backward()
            red.check_views()
            red.all_reduce()
            opt.step()
        w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        q.put((rank, w0.numpy().copy(), w.numpy().copy(), idx))  # plain arrays: no shared-memory handles
    finally:
        dist.destroy_process_group()


def _single_process_reference(w0):
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 3, padding=1))
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(w0[off:off + p.numel()].view_as(p))
            off += p.numel()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 2, 8, 8, generator=g)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x_all), y_all).backward()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


@pytest.mark.timeout(180)
def test_two_rank_data_parallel_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, w0a, wa, ia), (_, w0b, wb, ib) = [(r, torch.from_numpy(a), torch.from_numpy(b), i) for r, a, b, i in res]
    assert torch.equal(w0a, w0b), "broadcast_parameters did not synchronise the replicas"
    assert torch.equal(wa, wb), "replicas diverged after all-reduced steps"
    assert sorted(ia + ib) == [0, 1, 2, 3] and not set(ia) & set(ib)
    # mean of per-shard mean losses == full-batch mean loss (equal shard sizes): same as one process
    ref = _single_process_reference(w0a)
    assert torch.allclose(wa, ref, atol=1e-6, rtol=1e-5)


def test_flat_bucket_views_and_single_rank_noop():
    from superresolution_amd.dp import FlatGradAllReducer
    lin = torch.nn.Linear(3, 2)
    red = FlatGradAllReducer(lin.parameters())
    lin(torch.ones(1, 3)).sum().backward()
    red.check_views()
    assert red.flat.abs().sum() > 0
    before = red.flat.clone()
    red.all_reduce()
    assert torch.equal(before, red.flat)
    lin.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        red.check_views()


def test_gather_mode_fills_the_same_bucket():
    """gather=True: backward assigns the gradients, all_reduce() moves them into the bucket in one copy; the bucket,
    the optimiser's view of it and a second accumulation behave like the in-place mode."""
    from superresolution_amd.dp import FlatGradAllReducer
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 3))
    net_b = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 3))
    net_b.load_state_dict(net_a.state_dict())
    unused = torch.nn.Parameter(torch.ones(2))
    ra = FlatGradAllReducer(list(net_a.parameters()))
    rb = FlatGradAllReducer(list(net_b.parameters()) + [unused], gather=True)
    x = torch.randn(6, 5)
    for _ in range(2):                      # the second round checks zero_grad / re-pointing
        ra.zero_grad(); rb.zero_grad()
        assert all(p.grad is None for p in rb.params)
        net_a(x).square().sum().backward()
        net_b(x).square().sum().backward()
        net_b(2 * x).sum().backward()       # accumulation over two backward passes before the collect
        net_a(2 * x).sum().backward()
        ra.all_reduce(); rb.all_reduce()
        rb.check_views()
        n = ra.flat.numel()
        assert torch.allclose(rb.flat[:n], ra.flat, rtol=1e-6, atol=1e-7)
        assert torch.equal(rb.flat[n:], torch.zeros(2))
        rb.all_reduce()                     # idempotent: nothing left to collect
        assert torch.allclose(rb.flat[:n], ra.flat, rtol=1e-6, atol=1e-7)


def _bucket_worker(rank, world, port, q):
    """Two ranks, the same gradients reduced three ways: one bucket after backward, three buckets started from inside
    the backward pass in reverse parameter order (the way the default network's hand-written backward announces its
    weight gradients), and three buckets all started after backward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (4, 4, 3), (9,), (2, 8), (6,))]
        versions = [p._version for p in params]
        holder = torch.nn.ParameterList(params)
        broadcast_parameters(holder, src=0)
        bumped = all(p._version > v for p, v in zip(params, versions))   # packed-weight caches key on the version
        g = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(p.shape, generator=g) for p in params]
        out = []
        for buckets, announce in ((1, False), (3, True), (3, False)):
            red = FlatGradAllReducer(params, buckets=buckets)
            red.zero_grad()
            for p, gr in reversed(list(zip(params, grads))):           # backward order = reverse parameter order
                red.grad_view(p).add_(gr)
                if announce:
                    red.mark_ready(p)
            started_in_backward = list(red.launch_log)
            red.all_reduce()
            out.append((red.flat.clone().numpy(), red.n_buckets, started_in_backward, list(red.launch_log)))
        q.put((rank, bumped, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_overlapped_all_reduce_is_bit_equal_to_one_bucket():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, bumped, out in res:
        assert bumped, "broadcast_parameters must bump the parameters' version counters"
        (one, nb1, _, _), (over, nb3, started, log3), (late, _, started_late, log_late) = out
        assert nb1 == 1 and nb3 == 3
        assert started == [2, 1, 0], "buckets must start as backward completes them, last parameters first"
        assert started_late == [] and log_late == [2, 1, 0]
        assert (one == over).all() and (one == late).all(), "bucketing changed the reduced gradients"
    assert (res[0][2][0][0] == res[1][2][0][0]).all(), "ranks disagree after the all-reduce"


def test_bucket_layout_covers_the_buffer_in_whole_parameters():
    from superresolution_amd.dp import FlatGradAllReducer
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 1, 30, 5, 5, 49)]
    red = FlatGradAllReducer(params, buckets=4)
    assert red.bucket_range[0][0] == 0 and red.bucket_range[-1][1] == red.flat.numel()
    for (lo, hi), (lo2, _hi2) in zip(red.bucket_range, red.bucket_range[1:]):
        assert hi == lo2 and hi > lo
    assert sorted(set(red.bucket_of)) == list(range(red.n_buckets)) and red.bucket_of == sorted(red.bucket_of)
    # more buckets than parameters: never an empty bucket
    red = FlatGradAllReducer(params[:2], buckets=8)
    assert red.n_buckets <= 2 and all(hi > lo for lo, hi in red.bucket_range)


def _rearm_worker(rank, world, port, q):
    """Two ranks, three steps, gradients announced from 'inside backward' (mark_ready); the buffer is zeroed WITHOUT
    FlatGradAllReducer.zero_grad (flat.zero_() / optimizer.zero_grad(set_to_none=False)).  Every step must still run
    its collectives: the reduced buffer equals the mean of the two ranks' gradients each time."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd.dp import FlatGradAllReducer
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (4, 4, 3), (9,))]
        red = FlatGradAllReducer(params, buckets=2)
        opt = torch.optim.SGD(params, lr=0.0)
        out = []
        for step in range(3):
            if step == 1:
                red.flat.zero_()
            elif step == 2:
                opt.zero_grad(set_to_none=False)
            g = torch.Generator().manual_seed(1000 * step + rank)
            grads = [torch.randn(p.shape, generator=g) for p in params]
            red.begin_backward()
            for p, gr in reversed(list(zip(params, grads))):
                red.grad_view(p).add_(gr)
                red.mark_ready(p)
            in_backward = list(red.launch_log)
            red.all_reduce()
            red.check_views()
            local = torch.cat([gr.reshape(-1) for gr in grads])
            out.append((red.flat.clone().numpy(), local.numpy(), in_backward))
        # two backward passes in one step: the first under no_sync() starts nothing, the second reduces the sum
        red.zero_grad()
        ga = [torch.full(p.shape, float(rank + 1)) for p in params]
        gb = [torch.full(p.shape, 10.0 * (rank + 1)) for p in params]
        with red.no_sync():
            red.begin_backward()
            for p, gr in reversed(list(zip(params, ga))):
                red.grad_view(p).add_(gr)
                red.mark_ready(p)
            deferred = list(red.launch_log)
        red.begin_backward()
        for p, gr in reversed(list(zip(params, gb))):
            red.grad_view(p).add_(gr)
            red.mark_ready(p)
        second = list(red.launch_log)
        # a third backward now would race the collectives already started: refused
        try:
            red.begin_backward()
            refused = False
        except RuntimeError:
            refused = True
        red.all_reduce()
        q.put((rank, out, deferred, second, refused, red.flat.clone().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_buckets_rearm_without_zero_grad_and_accumulation_under_no_sync():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rearm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, out0, def0, sec0, ref0, acc0), (_, out1, def1, sec1, ref1, acc1) = res
    for step in range(3):
        red0, loc0, log0 = out0[step]
        red1, loc1, log1 = out1[step]
        assert log0 == [1, 0] and log1 == [1, 0], f"step {step}: the buckets did not start from inside backward"
        assert (red0 == red1).all(), f"step {step}: replicas hold different reduced gradients"
        assert np.allclose(red0, 0.5 * (loc0 + loc1), rtol=1e-6, atol=1e-7), f"step {step}: no collective ran"
    assert def0 == [] and def1 == [], "no_sync() must not start a collective"
    assert sec0 == [1, 0] and sec1 == [1, 0]
    assert ref0 and ref1, "a backward pass beside running collectives must be refused"
    assert (acc0 == acc1).all() and np.allclose(acc0, 0.5 * (11.0 + 22.0))


def test_single_rank_second_backward_is_not_refused():
    """Without a collective nothing is in flight: two announced backward passes in a row simply accumulate."""
    from superresolution_amd.dp import FlatGradAllReducer
    params = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))]
    red = FlatGradAllReducer(params, buckets=2)
    for _ in range(2):
        red.begin_backward()
        for p in reversed(params):
            red.grad_view(p).add_(1.0)
            red.mark_ready(p)
    red.all_reduce()
    assert torch.equal(red.flat, torch.full((5,), 2.0))


def _hook_worker(rank, world, port, q):
    """Gather mode with post-accumulate hooks (the temporal network's and RepRFN's reducer): buckets start from inside
    torch autograd's backward, last layers first, and the reduced buffer equals the plain gather mode's bit for bit;
    a parameter that receives no gradient is picked up by all_reduce(); accumulation runs under no_sync()."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
        torch.manual_seed(9)
        def make():
            torch.manual_seed(9)
            return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                                       torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
        g = torch.Generator().manual_seed(50 + rank)
        x, y = torch.randn(2, 3, 8, 8, generator=g), torch.randn(2, 2, 8, 8, generator=g)
        out = {}
        for mode in ("plain", "hooks"):
            net = make()
            unused = torch.nn.Parameter(torch.ones(3))
            broadcast_parameters(net, src=0)
            red = FlatGradAllReducer(list(net.parameters()) + [unused], gather=True, buckets=3)
            if mode == "hooks":
                red.attach_hooks()
            logs = []
            for step in range(2):
                red.zero_grad()
                torch.nn.functional.mse_loss(net(x), y).backward()
                in_backward = list(red.launch_log)
                red.all_reduce()
                red.check_views()
                logs.append((in_backward, list(red.launch_log)))
            # two backward passes in one step
            red.zero_grad()
            with red.no_sync():
                torch.nn.functional.mse_loss(net(x), y).backward()
                deferred = list(red.launch_log)
            torch.nn.functional.mse_loss(net(2 * x), y).backward()
            red.all_reduce()
            out[mode] = (red.flat.clone().numpy(), logs, deferred, red.n_buckets)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_hooked_gather_mode_overlaps_and_equals_plain_gather_mode():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hook_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, out in res:
        flat_p, logs_p, _, nb = out["plain"]
        flat_h, logs_h, deferred, _ = out["hooks"]
        assert nb == 3
        assert (flat_p == flat_h).all(), "hooked buckets changed the reduced gradients"
        for in_bw, after in logs_p:
            assert in_bw == [] and after == [2, 1, 0]
        for in_bw, after in logs_h:
            # the last bucket holds the unused parameter: only all_reduce() can complete it; the others start in backward
            assert in_bw == [1, 0] and after == [1, 0, 2], (in_bw, after)
        assert deferred == []
    assert (res[0][1]["hooks"][0] == res[1][1]["hooks"][0]).all()
