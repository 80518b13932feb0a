"""torch.ops.pevae_amd.* (pevae_amd/torch_ops.py): registration and fake (meta) shapes without a GPU; on a GPU the operators
return what pevae_amd.ops returns."""
import pytest
import torch


def test_operators_are_registered_with_fake_shapes():
    import pevae_amd.torch_ops  # noqa: F401
    hid = torch.empty(12, 64, device="meta")
    w, b = torch.empty(500, 64, device="meta"), torch.empty(500, device="meta")
    ids, lp = torch.ops.pevae_amd.vocab_topk(hid, w, b, 7, 1.0, -1)
    assert ids.shape == (12, 7) and ids.dtype == torch.int64 and lp.shape == (12, 7) and lp.dtype == torch.float32
    ids, lp = torch.ops.pevae_amd.vocab_argmax(hid, w, b, 1.0, 3)
    assert ids.shape == (12,) and lp.shape == (12,)
    ids, lp = torch.ops.pevae_amd.vocab_sample(hid, w, b, 1.0, 3, 5)
    assert ids.shape == (12,) and lp.dtype == torch.float32
    h0 = torch.empty(4, 64, device="meta")
    seq, lp, hn = torch.ops.pevae_amd.greedy_decode(torch.empty(500, 16, device="meta"), torch.empty(192, 16, device="meta"),
                                                    torch.empty(192, 64, device="meta"), torch.empty(192, device="meta"),
                                                    torch.empty(192, device="meta"), w, b, h0, 9, 1, 1.0, -1)
    assert seq.shape == (4, 9) and lp.shape == (4, 9) and hn.shape == (4, 64)


def test_operators_refuse_cpu_tensors():
    import pevae_amd.torch_ops  # noqa: F401
    with pytest.raises(Exception):
        torch.ops.pevae_amd.vocab_argmax(torch.zeros(2, 64), torch.zeros(10, 64), torch.zeros(10), 1.0, -1)


@pytest.mark.gpu
def test_operators_match_the_package_wrappers():
    import pevae_amd.torch_ops  # noqa: F401
    from pevae_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    hid = torch.randn(33, 128, generator=g).to(dev)
    w, b = (torch.randn(1000, 128, generator=g) * 0.1).to(dev), torch.randn(1000, generator=g).to(dev)
    a = torch.ops.pevae_amd.vocab_topk(hid, w, b, 5, 0.9, 4)
    c = ops.vocab_topk(hid, w, b, 5, 0.9, 4)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    a = torch.ops.pevae_amd.vocab_argmax(hid, w, b, 0.9, 4)
    c = ops.vocab_argmax(hid, w, b, 0.9, 4)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    assert torch.equal(a[0], torch.ops.pevae_amd.vocab_topk(hid, w, b, 1, 0.9, 4)[0][:, 0])
