"""Per-shape timing of the training weight preparation (svc_conv_weight_prep_f32: weight-norm row norms + the forward and
dgrad operand packs of one convolution) and of its adjoint (svc_conv_weight_grad_f32) on the training step's largest weights.
    python scripts/bench_weight_prep.py            (on a GPU box; prints one row per shape: us, GB/s moved)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import svc_hip as S  # noqa: E402

SHAPES = [  # (label, kind, R, C2, K, s, shift, Kd)
    ("DiscP 1024->1024 k5", 0, 1024, 1024, 5, 1, 0, None),
    ("DiscP 512->1024 k5 s3", 1, 1024, 512, 5, 3, 0, 2),
    ("DiscP 128->512 k5 s3", 1, 512, 128, 5, 3, 0, 2),
    ("DiscS 1024->1024 k5", 0, 1024, 1024, 5, 1, 0, None),
    ("DiscS g256 4->1024 k41", 0, 1024, 4, 41, 1, 0, None),
    ("WN in 192->384 k5", 0, 384, 192, 5, 1, 0, None),
    ("FFN 192->768 k3", 0, 768, 192, 3, 1, 0, None),
    ("FFN 768->192 k3", 0, 192, 768, 3, 1, 0, None),
    ("res 256 k11", 0, 256, 256, 11, 1, 0, None),
    ("res 128 k11", 0, 128, 128, 11, 1, 0, None),
    ("res 32 k7", 0, 32, 32, 7, 1, 0, None),
    ("ups 512->256 k16 s8", 2, 512, 256, 16, 8, 0, None),
    ("ups 256->128 k16 s8", 2, 256, 128, 16, 8, 0, None),
]


def time_us(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':28s} {'blocks':>6s} {'prep us':>9s} {'GB/s':>7s} {'grad us':>9s} {'GB/s':>7s}")
    tot_p = tot_g = 0.0
    for label, kind, R, C2, K, s, shift, Kd in SHAPES:
        pl = S.ConvWeightPlan(kind, R, C2, K, s=s, shift=shift, Kd=Kd)
        v = torch.randn(R, C2, K, device=dev)
        g = torch.rand(R, 1, 1, device=dev) + 0.5
        pl.prepare(v, g)
        dwd = torch.randn(pl.Od, pl.Id, pl.Kd, device=dev)
        tp = time_us(lambda: pl.prepare(v, g))
        tg = time_us(lambda: pl.grad(v, g, dwd))
        bytes_p = 4.0 * (2 * v.numel() + 2 * v.numel())          # read twice (norm, scatter), written to wp and wt
        bytes_g = 4.0 * 3 * v.numel()
        nb = S.tlib().svc_conv_weight_prep_blocks(R, C2, K)
        print(f"{label:28s} {nb:6d} {tp:9.1f} {bytes_p / tp / 1e3:7.0f} {tg:9.1f} {bytes_g / tg / 1e3:7.0f}")
        tot_p += tp
        tot_g += tg
    print(f"sum prep {tot_p:.0f} us, grad {tot_g:.0f} us")


if __name__ == "__main__":
    main()
