"""CPU tests of host-side helpers added in round 5 (pure torch / Python, no device calls): the padded grouping of unequal chunks for the
unit encoder, and the autograd bookkeeping ops whose backward replaces torch's slice / stack backward passes."""
import torch


def test_batch_padded_groups_by_padding_waste_and_restores_order():
    """vencoder.encoder.batch_padded: waves sorted by length, a group takes items down to 75 % of its longest, every item comes back
    cut to ITS frames in the input order (Svc.slice_inference's chunks, reference inference/infer_tool.py:446-495)."""
    from vencoder.encoder import batch_padded
    lens = [16000, 9000, 15999, 401, 12000, 16000, 11999]
    wavs = [torch.full((n,), float(i)) for i, n in enumerate(lens)]
    calls = []

    def run(x, lengths):
        calls.append(list(lengths))
        assert x.shape == (len(lengths), 1, max(lengths))
        for b, n in enumerate(lengths):
            assert float(x[b, 0, n:].abs().sum()) == 0.0                        # zero padded
        frames = [n // 320 for n in lengths]
        y = torch.zeros(len(lengths), 4, max(frames))
        for b, n in enumerate(lengths):
            y[b, :, :frames[b]] = x[b, 0, 0]                                     # tag every frame with the item's id
        return y, frames
    out = batch_padded(wavs, run)
    assert calls == [[16000, 16000, 15999, 12000], [11999, 9000], [401]]
    for i, (n, o) in enumerate(zip(lens, out)):
        assert o.shape == (1, 4, n // 320) and bool((o == float(i)).all())
    # stereo input is averaged, as the encoders' own `encoder` does
    st = batch_padded([torch.stack([torch.ones(800), 3 * torch.ones(800)], 1)], lambda x, l: (x.mean(2, keepdim=True).expand(-1, 2, 5), [5]))
    assert float(st[0].mean()) == 2.0


def test_chunk_stack_and_split_ops_match_plain_torch_autograd():
    """svc_autograd.chunk_channels(views=True) / stack_qkv / split_batch against the Python slices, torch.stack and batch slices they
    replace: same forward values, same gradients (the ops only change HOW the backward assembles them)."""
    import svc_autograd as A
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 12, 5, generator=g, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    w = [torch.randn(2, 4, 5, generator=g) for _ in range(3)]
    outs = A.chunk_channels(x, 3, views=True)
    assert all(torch.equal(o, xr[:, 4 * i:4 * (i + 1)]) for i, o in enumerate(outs))
    sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
    sum((xr[:, 4 * i:4 * (i + 1)] * wi).sum() for i, wi in enumerate(w)).backward()
    assert torch.equal(x.grad, xr.grad)
    H, dk, C = 2, 3, 5
    ps = [torch.randn(H * dk, C, 1, generator=g, requires_grad=True) for _ in range(3)]
    pr = [p.detach().clone().requires_grad_(True) for p in ps]
    fused = A.stack_qkv(*ps, H)
    ref = torch.stack([p.view(H, dk, C) for p in pr], 1).reshape(3 * H * dk, C, 1)
    assert torch.equal(fused, ref)
    gw = torch.randn(fused.shape, generator=g)
    fused.backward(gw)
    ref.backward(gw)
    assert all(torch.equal(a.grad, b.grad) and a.grad.is_contiguous() for a, b in zip(ps, pr))
    y = torch.randn(6, 1, 7, generator=g, requires_grad=True)
    yr = y.detach().clone().requires_grad_(True)
    a, b = A.split_batch(y, 4)
    (a.pow(2).sum() + 3 * b.sum()).backward()
    (yr[:4].pow(2).sum() + 3 * yr[4:].sum()).backward()
    assert torch.equal(y.grad, yr.grad)
    a2, _ = A.split_batch(y.detach().clone().requires_grad_(True), 4)          # one half unused: its gradient is zero
    a2.sum().backward()


def test_precision_mode_switches_are_host_logic():
    """SynthesizerTrn.half() / split_f16() / float() only choose the generator's pipeline (no GPU needed to flip them): which
    generators accept which mode, and that a mode switch drops captured graphs."""
    import pytest
    import models
    from oracle import weights as W

    def build(cfg):
        kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
        return models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw).eval()

    net = build(W.full_config())
    net._graphs["x"] = object()
    assert net.split_f16() is net and net.dec.half_mode == "split" and not net._graphs
    assert next(net.parameters()).dtype == torch.float32
    net.half()
    assert net.dec.half_mode is True
    net.split_f16(False)
    assert net.dec.half_mode is False
    net.split_f16()
    net.float()
    assert net.dec.half_mode is False
    snake = W.full_config()
    snake["vocoder_name"] = "nsf-snake-hifigan"
    net = build(snake)
    net.half()                                        # the snake generator has a 16-bit form ...
    assert net.dec.half_mode is True
    net.split_f16()                                   # ... and a split one
    assert net.dec.half_mode == "split"
    net = build(W.small_config())                     # stage widths 64 / 32 / 16 / 8 / 4: zero-padded to multiples of 16 in both forms
    net.half()
    assert net.dec.half_mode is True
    odd = W.small_config()
    odd["resblock_kernel_sizes"] = [3, 9, 11]         # a tap count without a 16-bit kernel: refused, the model stays fp32
    net = build(odd)
    for switch in (net.half, net.split_f16):
        with pytest.raises(NotImplementedError):
            switch()
    assert net.dec.half_mode is False
