"""The fused coupling kernel (csrc/flow_fused.hip, svc_coupling_fused_h): ONE launch per ResidualCouplingLayer of the flow in the
16-bit and the split inference modes (reference modules/modules.py:110-138,288-307, models.py:45-52 with the Flip folded into a
negative channel stride).  Checked against the oracle's flow (float64 here: the exact answer), with the fp32 kernels' own error
beside it: the split planes must be fp32-level, the fp16 planes within the reference's half-mode class; both directions, padding
masks, tile borders and sequences shorter than a halo, a per-frame conditioning tensor, and bit-equality of hipGraph replay."""
import pytest
import torch

from oracle import svc_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu


def _flow_and_sd(dev, seed=5):
    import models
    cfg = W.full_config()
    sd = {k: v for k, v in W.make_state_dict(cfg, seed).items() if k.startswith("flow.")}
    flow = models.ResidualCouplingBlock(cfg["inter_channels"], cfg["hidden_channels"], 5, 1, cfg.get("n_flow_layer", 4),
                                        gin_channels=cfg["gin_channels"], share_parameter=cfg.get("flow_share_parameter", False))
    flow.load_state_dict({k[len("flow."):]: v for k, v in sd.items()})
    return cfg, flow.to(dev).eval(), sd


def _inputs(B, T, lengths, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 192, T, generator=g)
    spk = torch.randn(B, 768, 1, generator=g)
    lens = torch.tensor(lengths)
    mask = (torch.arange(T).view(1, 1, T) < lens.view(B, 1, 1)).float()
    return x * mask, mask, spk


def _exact(sd, cfg, x, mask, spk, reverse):
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        return O.flow(x.double(), mask.double(), spk.double(), sd64, cfg, reverse)


@pytest.mark.parametrize("B,T,lengths", [(1, 862, [862]), (2, 131, [131, 77]), (1, 48, [48]), (1, 49, [49]), (3, 7, [7, 3, 1]),
                                         (1, 200, [150])])
@pytest.mark.parametrize("reverse", [True, False])
def test_fused_coupling_matches_the_flow_in_float64(dev, B, T, lengths, reverse):
    cfg, flow, sd = _flow_and_sd(dev)
    assert flow._fusable()
    x, mask, spk = _inputs(B, T, lengths, seed=B * 100 + T)
    exact = _exact(sd, cfg, x, mask, spk, reverse)
    scale = exact.abs().max().item()
    run = lambda: flow(x.to(dev), mask.to(dev), g=spk.to(dev), reverse=reverse)
    with torch.no_grad():
        y32 = run()
        y32 = (y32[0] if isinstance(y32, tuple) else y32).cpu()
        flow.set_half(True, split=True)
        assert flow.fused_mode == "split"
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        import svc_hip as S
        S.hl_range_flag(flag)
        try:
            ys = run()
        finally:
            S.hl_range_flag(None)
        ys = (ys[0] if isinstance(ys, tuple) else ys).cpu()
        flow.set_half(True)
        assert flow.fused_mode is True
        yh = run()
        yh = (yh[0] if isinstance(yh, tuple) else yh).cpu()
        flow.set_half(False)
        assert torch.equal((lambda r: r[0] if isinstance(r, tuple) else r)(run()).cpu(), y32)
    e32 = (y32.double() - exact).abs().max().item() / scale
    es = (ys.double() - exact).abs().max().item() / scale
    eh = (yh.double() - exact).abs().max().item() / scale
    print(f"flow B={B} T={T} reverse={reverse}: fp32 kernels {e32:.2e}, fused split {es:.2e}, fused fp16 {eh:.2e} (of max |exact| {scale:.3g})")
    assert int(flag.item()) == 0
    assert ys.shape == exact.shape and es < 3e-6 and es < 6 * e32 + 3e-7, (es, e32)
    assert eh < 2e-2, eh
    # outside the mask the flow's state is zero (x * mask at every update), as in the reference
    assert (ys * (1 - mask)).abs().max().item() == 0.0 and (yh * (1 - mask)).abs().max().item() == 0.0


def test_fused_coupling_with_per_frame_conditioning_and_graph_replay(dev):
    """Speaker mix hands the flow a conditioning tensor per frame (models.py:505-509: g [B, gin, T]); and a captured flow replays
    bit-identically (the kernel holds no state besides its arguments)."""
    cfg, flow, sd = _flow_and_sd(dev, seed=9)
    B, T = 1, 300
    x, mask, _ = _inputs(B, T, [T], seed=3)
    spk = torch.randn(B, 768, T, generator=torch.Generator().manual_seed(4))
    exact = _exact(sd, cfg, x, mask, spk, True)
    xd, md, sd_ = x.to(dev), mask.to(dev), spk.to(dev)
    with torch.no_grad():
        flow.set_half(True, split=True)
        y = flow(xd, md, g=sd_, reverse=True)
        assert (y.cpu().double() - exact).abs().max().item() / exact.abs().max().item() < 3e-6
        flow.set_half(True)
        y16 = flow(xd, md, g=sd_, reverse=True)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            flow(xd, md, g=sd_, reverse=True)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            yg = flow(xd, md, g=sd_, reverse=True)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, y16)


def test_flow_blocks_without_a_fused_form_keep_the_fp32_launches(dev):
    import models
    small = W.small_config()
    flow = models.ResidualCouplingBlock(small["inter_channels"], small["hidden_channels"], 5, 1, 4, gin_channels=small["gin_channels"])
    if small["hidden_channels"] != 192:
        assert not flow._fusable()
        flow.set_half(True)
        assert flow.fused_mode is False
