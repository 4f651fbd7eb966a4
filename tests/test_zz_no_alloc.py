"""The allocation-free entry point, called through the C ABI: srhip_layernorm_cf_bwd_f32_static must give exactly
what srhip_layernorm_cf_bwd_f32 (caller-owned workspace) gives, also on a second, smaller call that finds the first
call's contents in the library's fixed scratch array, and must refuse shapes beyond that array."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(n, c, h, w, seed):
    from superresolution_amd._lib import check, lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g).cuda()
    dy = torch.randn(n, c, h, w, generator=g).cuda()
    wt = torch.randn(c, generator=g).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for static in (False, True):
        dx = torch.full_like(x, float("nan"))
        dw = torch.full((c,), float("nan"), device="cuda")
        db = torch.full((c,), float("nan"), device="cuda")
        if static:
            check(lib.srhip_layernorm_cf_bwd_f32_static(x.data_ptr(), wt.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                        db.data_ptr(), n, c, h * w, 1e-6, stream))
        else:
            ws = torch.empty((lib.srhip_layernorm_cf_bwd_workspace_bytes(n, c, h * w),), dtype=torch.uint8, device="cuda")
            check(lib.srhip_layernorm_cf_bwd_f32(x.data_ptr(), wt.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                 db.data_ptr(), ws.data_ptr(), n, c, h * w, 1e-6, stream))
        torch.cuda.synchronize()
        outs.append((dx, dw, db))
    return outs


def test_static_scratch_entry_point_matches_the_workspace_form_twice():
    for shape in ((2, 60, 96, 128), (1, 24, 17, 29)):      # the second call is smaller and meets a used scratch array
        (dx0, dw0, db0), (dx1, dw1, db1) = _both(*shape, seed=sum(shape))
        assert bool(torch.isfinite(dx1).all()) and bool(torch.isfinite(dw1).all()) and bool(torch.isfinite(db1).all())
        assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)


def test_static_scratch_entry_point_rejects_shapes_beyond_its_array():
    from superresolution_amd._lib import lib
    t = torch.zeros(8, device="cuda")
    p = t.data_ptr()
    assert lib.srhip_layernorm_cf_bwd_f32_static(p, p, p, p, p, p, 2, 4, 1 << 20, 1e-6, None) == -1   # 2 * 2^20 pixels
    assert b"exceed the fixed scratch" in lib.srhip_last_error()
    assert lib.srhip_layernorm_cf_bwd_f32_static(p, p, p, p, p, p, 1, 257, 64, 1e-6, None) == -1       # 257 channels


def _static_call(x, wt, dy, dx, dw, db, stream):
    from superresolution_amd._lib import check, lib
    n, c, h, w = x.shape
    check(lib.srhip_layernorm_cf_bwd_f32_static(x.data_ptr(), wt.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                db.data_ptr(), n, c, h * w, 1e-6, stream))


def test_static_scratch_entry_point_allocates_nothing_and_replays_from_a_graph():
    """What the entry point exists for: no device allocation happens in a call (neither through torch's allocator nor
    behind it), and after one warm-up call outside capture (which resolves the scratch array's address on this device)
    the call can be recorded into a HIP graph and replayed."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 40, 33, 65, generator=g).cuda()
    dy = torch.randn(1, 40, 33, 65, generator=g).cuda()
    wt = torch.randn(40, generator=g).cuda()
    outs = [torch.empty_like(x), torch.empty(40, device="cuda"), torch.empty(40, device="cuda")]
    _static_call(x, wt, dy, *outs, torch.cuda.current_stream().cuda_stream)        # warm-up: address lookup, code load
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs]
    alloc0, reserved0, free0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), torch.cuda.mem_get_info()[0]
    _static_call(x, wt, dy, *outs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == alloc0 and torch.cuda.memory_reserved() == reserved0
    assert torch.cuda.mem_get_info()[0] == free0, "device memory was allocated behind torch's allocator"
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)
    # recorded once, replayed on fresh inputs written into the same buffers
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _static_call(x, wt, dy, *outs, torch.cuda.current_stream().cuda_stream)
    x2 = torch.randn(1, 40, 33, 65, generator=g).cuda()
    dy2 = torch.randn(1, 40, 33, 65, generator=g).cuda()
    x.copy_(x2); dy.copy_(dy2)
    for t in outs:
        t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in outs]
    _static_call(x, wt, dy, *outs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for a, b in zip(replayed, outs):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
