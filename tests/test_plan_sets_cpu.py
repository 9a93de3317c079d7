"""CPU suite: the bracket / token logic of svc_hip.PlanSets (one weight preparation per forward pass), with the C-ABI calls
replaced by recorders — what launches when, which plans are served from the bracket's launch, what `leave` withdraws, and that
a changed parameter storage sends the set back to recording.  The kernels themselves are checked on the GPU
(tests/test_train_ops_gpu.py::test_plan_sets_one_launch_equals_per_plan_preparation)."""
import torch

import svc_hip as S


class _FakeLib:
    def __init__(self):
        self.calls = []

    def svc_conv_weight_prep_f32(self, args, stream):
        self.calls.append("single")
        return 0

    def svc_conv_weight_prep_multi_f32(self, host, dev, rows, blocks, n, stream):
        self.calls.append(("multi", n))
        return 0

    def svc_conv_weight_prep_blocks(self, R, C2, K):
        return ((R + 31) // 32) * max(1, (C2 * K + 1023) // 1024)


def _patch(monkeypatch):
    lib = _FakeLib()
    monkeypatch.setattr(S, "tlib", lambda: lib)
    monkeypatch.setattr(S, "require_gpu", lambda *a: None)
    monkeypatch.setattr(S, "stream_ptr", lambda: 0)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    return lib


def test_plan_sets_bracket_semantics(monkeypatch):
    lib = _patch(monkeypatch)
    P = S.ConvWeightPlan
    plans = [P(P.DENSE, 8, 4, 3), P(P.DENSE, 6, 2, 1)]
    vs = [torch.nn.Parameter(torch.randn(8, 4, 3)), torch.nn.Parameter(torch.randn(6, 2, 1))]
    gs = [torch.nn.Parameter(torch.rand(8, 1, 1) + 0.5), None]
    other = torch.randn(6, 2, 1)                       # a computed weight (e.g. spectral norm): not parameter storage

    def forward():
        for pl, v, g in zip(plans, vs, gs):
            pl.prepare(v, g)

    sets = S.PlanSets()
    params = [p for p in vs + gs if p is not None]
    # 1st bracket records: per-plan launches, a set of two plans afterwards
    sets.enter("f", params)
    forward()
    plans[1].prepare(other)                            # not recorded: its storage is no parameter's
    sets.leave("f")
    assert lib.calls == ["single", "single", "single"]
    assert [e[0] for e in sets.sets["f"]["items"]] == plans
    assert S.PlanSets.recording is None
    # 2nd bracket: ONE multi call, the plans' own prepare() calls are served from it — also when called twice
    lib.calls.clear()
    sets.enter("f", params)
    forward()
    forward()
    assert lib.calls == [("multi", 2)]
    plans[1].prepare(other)                            # a different tensor still prepares by itself
    assert lib.calls == [("multi", 2), "single"]
    sets.leave("f")
    # outside the bracket the tokens are gone: every prepare launches
    lib.calls.clear()
    forward()
    assert lib.calls == ["single", "single"]
    # a parameter that moved (new storage) invalidates the set: the next bracket records again
    vs[0] = torch.nn.Parameter(vs[0].detach().clone())
    params = [p for p in vs + gs if p is not None]
    lib.calls.clear()
    sets.enter("f", params)
    forward()
    sets.leave("f")
    assert lib.calls == ["single", "single"]
    lib.calls.clear()
    sets.enter("f", params)
    forward()
    sets.leave("f")
    assert lib.calls == [("multi", 2)]


def test_plan_sets_disabled_is_transparent(monkeypatch):
    lib = _patch(monkeypatch)
    P = S.ConvWeightPlan
    pl, v = P(P.DENSE, 4, 4, 1), torch.nn.Parameter(torch.randn(4, 4, 1))
    sets = S.PlanSets()
    sets.enabled = False
    for _ in range(3):
        sets.enter("f", [v])
        pl.prepare(v, None)
        sets.leave("f")
    assert lib.calls == ["single"] * 3 and not sets.sets
