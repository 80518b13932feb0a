"""pv_vocab_argmax_noalloc (the entry point that allocates nothing: per-slab partials in a device array built into the
library) against pv_vocab_argmax with a caller workspace, both through the C ABI: symbols bit-exact, log-probabilities
bit-exact (the same kernels on the same operands), on two calls in a row, the second with a smaller input -- the built-in
array keeps the first call's contents, and nothing of them may leak into the second result.  Also: sizes beyond the
built-in array are rejected with PV_ERR_LIMIT, and the call makes no device allocation (torch's allocator statistics and
hipMemGetInfo stay put)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _operands(Tn, V, H, seed, dev):
    gen = torch.Generator().manual_seed(seed)
    hid = (torch.randn(Tn, H, generator=gen) * 0.5).to(torch.bfloat16).to(dev).contiguous()
    W = (torch.randn(V, H, generator=gen) * 0.05).to(torch.bfloat16).to(dev).contiguous()
    b = (torch.randn(V, generator=gen) * 0.1).to(dev).contiguous()
    return hid, W, b


def _both(Tn, V, H, seed, mask, dev):
    from pevae_amd import _lib
    lib = _lib.load()
    hid, W, b = _operands(Tn, V, H, seed, dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pv_vocab_argmax_ws_bytes(Tn, V), dtype=torch.uint8, device=dev)
    ids_a = torch.full((Tn,), -7, dtype=torch.int64, device=dev)
    lp_a = torch.full((Tn,), float("nan"), device=dev)
    ids_b, lp_b = ids_a.clone(), lp_a.clone()
    _lib.call("pv_vocab_argmax", hid.data_ptr(), H, W.data_ptr(), H, b.data_ptr(), Tn, V, H, 1.0 / 0.7, mask,
              ws.data_ptr(), ids_a.data_ptr(), lp_a.data_ptr(), st)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    alloc0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    _lib.call("pv_vocab_argmax_noalloc", hid.data_ptr(), H, W.data_ptr(), H, b.data_ptr(), Tn, V, H, 1.0 / 0.7, mask,
              ids_b.data_ptr(), lp_b.data_ptr(), st)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == alloc0
    assert free1 == free0, (free0, free1)
    # independent check of the workspace path itself (so both are right, not just equal)
    ref = torch.log_softmax((hid.double() @ W.double().t() + b.double()) / 0.7, -1)
    if mask >= 0:
        ref[:, mask] = -float("inf")
    want_lp, want_id = ref.max(-1)
    flips = ids_a != want_id
    if bool(flips.any()):
        assert float((want_lp - ref.gather(1, ids_a[:, None])[:, 0])[flips].max()) < 1e-3
    assert float((lp_a.double() - want_lp).abs().max()) < 2e-3
    return (ids_a, lp_a), (ids_b, lp_b)


def test_noalloc_argmax_matches_workspace_entry_point_twice():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from pevae_amd import _lib
    lib = _lib.load()
    assert lib.pv_vocab_argmax_noalloc_max_bytes() == 4 << 20
    # warm-up call on the device: looks the built-in array's address up (outside the allocation bracket below)
    (a, la), (b, lb) = _both(64, 1000, 64, 1, -1, dev)
    assert torch.equal(a, b) and torch.equal(la, lb)
    # call 1: the decode shape of the step (256 rows, V = 20004 with a ragged last slab), UNK masked
    (a, la), (b, lb) = _both(256, 20004, 512, 2, 1, dev)
    assert torch.equal(a, b) and torch.equal(la, lb)
    assert int(b.min()) >= 0 and int((b == 1).sum()) == 0
    # call 2, smaller in rows, vocabulary and width: stale partials of call 1 sit in the array beside its own
    (a, la), (b, lb) = _both(37, 5003, 128, 3, -1, dev)
    assert torch.equal(a, b) and torch.equal(la, lb)
    # call 3: back to a larger one
    (a, la), (b, lb) = _both(200, 20004, 512, 4, 3, dev)
    assert torch.equal(a, b) and torch.equal(la, lb)


def test_noalloc_argmax_rejects_sizes_beyond_the_builtin_array():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from pevae_amd import _lib
    lib = _lib.load()
    Tn, V, H = 1024, 20004, 64                      # 313 slabs x 1024 rows x 16 B = 5.1 MB > 4 MiB
    hid, W, b = _operands(Tn, V, H, 5, dev)
    ids = torch.empty(Tn, dtype=torch.int64, device=dev)
    lp = torch.empty(Tn, device=dev)
    rc = lib.pv_vocab_argmax_noalloc(hid.data_ptr(), H, W.data_ptr(), H, b.data_ptr(), Tn, V, H, C.c_float(1.0), -1,
                                     ids.data_ptr(), lp.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == -3 and b"exceeds the built-in scratch" in lib.pv_last_error()
