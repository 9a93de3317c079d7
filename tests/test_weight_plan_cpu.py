"""CPU suite: the index maps of svc_conv_weight_prep_f32 / svc_conv_weight_grad_f32 (csrc/train_ops.hip: wmap_index), restated
in Python from the header's definition, against the torch index reshapes the unfused path performs (svc_autograd.conv1d /
conv_transpose1d): forward operand [Id,Kd,OdP] and dgrad operand [Od,Kd,IdP] for dense, strided and transposed layers."""
import numpy as np
import pytest
import torch

import svc_autograd as A


def wmap(plan, r, c, k):
    """include/svc_hip.h, svc_conv_weight_args: (r, c, k) of the parameter -> (o, i, m) of the dense-conv weight."""
    if plan.kind == 0:
        return r, c, k
    if plan.kind == 1:
        kk = k + plan.shift
        m = kk // plan.s
        return r, (kk - m * plan.s) * plan.C2 + c, m
    mm = k // plan.s
    return (k - mm * plan.s) * plan.C2 + c, r, plan.Kd - 1 - mm


def emulate(plan, w):
    wp = np.zeros((plan.Id, plan.Kd, plan.OdP), np.float32)
    wt = np.zeros((plan.Od, plan.Kd, plan.IdP), np.float32)
    seen = set()
    for r in range(w.shape[0]):
        for c in range(w.shape[1]):
            for k in range(w.shape[2]):
                o, i, m = wmap(plan, r, c, k)
                assert 0 <= o < plan.Od and 0 <= i < plan.Id and 0 <= m < plan.Kd and (o, i, m) not in seen
                seen.add((o, i, m))
                wp[i, m, o] = w[r, c, k]
                wt[o, plan.Kd - 1 - m, i] = w[r, c, k]
    return wp, wt


def operands(wd, plan):
    Od, Id, Kd = wd.shape
    assert (Od, Id, Kd) == (plan.Od, plan.Id, plan.Kd)
    wp = np.zeros((Id, Kd, plan.OdP), np.float32)
    wp[:, :, :Od] = wd.permute(1, 2, 0).numpy()                 # svc_pack_conv1d_weight
    wt = np.zeros((Od, Kd, plan.IdP), np.float32)
    wt[:, :, :Id] = wd.flip(2).permute(0, 2, 1).numpy()         # svc_pack_conv1d_weight_T
    return wp, wt


@pytest.mark.parametrize("Cout,Cg,KS,s,pad", [(6, 3, 5, 3, 2), (4, 1, 128, 64, 32), (5, 1, 16, 8, 4), (3, 1, 8, 4, 2),
                                              (7, 1, 4, 2, 1), (3, 2, 5, 1, 2), (2, 2, 41, 4, 20), (4, 3, 7, 2, 3), (3, 2, 3, 2, 1)])
def test_conv_and_strided_conv_maps(Cout, Cg, KS, s, pad):
    torch.manual_seed(0)
    w = torch.randn(Cout, Cg, KS)
    plan = A.conv_plan(w.shape, s, pad)
    if s == 1:
        wd = w
    else:       # svc_autograd.conv1d's lowering of a stride-s conv (models.py:171-177, vdecoder/hifigan/models.py:343-348)
        KSd, shift, _ = A.strided_geometry(KS, s, pad)
        wd = torch.nn.functional.pad(w, (shift, s * KSd - KS - shift)).view(Cout, Cg, KSd, s).permute(0, 3, 1, 2) \
            .reshape(Cout, s * Cg, KSd)
    a, b = emulate(plan, w.numpy()), operands(wd, plan)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("Cin,Cout,KS,u", [(4, 3, 16, 8), (5, 2, 4, 2), (3, 3, 5, 2), (2, 4, 7, 3), (3, 2, 3, 1)])
def test_transposed_conv_map(Cin, Cout, KS, u):
    torch.manual_seed(1)
    w = torch.randn(Cin, Cout, KS)
    plan = A.conv_plan(w.shape, u, 0, transposed=True)
    M = (KS + u - 1) // u       # svc_autograd.conv_transpose1d's lowering (vdecoder/hifigan/models.py:340-342)
    wd = torch.nn.functional.pad(w, (0, M * u - KS)).view(Cin, Cout, M, u).flip(2).permute(3, 1, 0, 2).reshape(u * Cout, Cin, M)
    a, b = emulate(plan, w.numpy()), operands(wd, plan)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_loss_scaler_follows_gradscaler_rule():
    """optim.LossScaler = torch.cuda.amp.GradScaler's bookkeeping (train.py:143,192-213): skip + halve on a non-finite gradient of
    EITHER optimizer of an iteration, double after `growth_interval` clean iterations, 1/scale handed to the optimizer."""
    from optim import LossScaler

    class Opt:
        def __init__(self):
            self.finite, self.steps, self.grad_scale = True, 0, None

        def grads_finite(self):
            return self.finite

        def step(self):
            self.steps += 1

    sc = LossScaler(init_scale=1024.0, growth_interval=3)
    d, g = Opt(), Opt()
    assert sc.step(d) and sc.step(g) and d.grad_scale == 1 / 1024.0
    sc.update()
    assert sc.scale == 1024.0
    d.finite = False                                   # D overflows: its step is skipped, G's is not, the scale halves ONCE
    assert not sc.step(d) and sc.step(g)
    sc.update()
    assert sc.scale == 512.0 and d.steps == 1 and g.steps == 2 and sc.skipped == 1
    d.finite = True
    for i in range(3):                                 # three clean iterations (the counter restarted at the overflow) -> x2
        sc.step(d); sc.step(g); sc.update()
    assert sc.scale == 1024.0 and g.grad_scale == 1 / 512.0
    sc2 = LossScaler()
    sc2.load_state_dict(sc.state_dict())
    assert sc2.scale == sc.scale and LossScaler().scale == 65536.0
