"""The arithmetic behind the two operand-splitting modes, restated on the CPU with torch's own fp16 / bf16 conversions (round to nearest
even — what v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 do) and fp32 matmuls standing in for the matrix instruction's fp32 accumulation:

* split pipeline (csrc/conv1d_hl.hip): v = hi + lo in two fp16 pieces, product = hi*hi' + hi*lo' + lo*hi';
* three bf16 pieces, six products (v = p0 + p1 + p2, the piece products of weight >= 2^-16): the round-5 training mode SVC_MMA_BF16X6.
  It was as exact as fp32 and no faster, and was removed in round 6 (ABI 5); the arithmetic stays here as the comparison that
  explains why the inference pipeline splits into fp16 pieces and what range that costs.

What the kernels' headers claim is checked here without a GPU: the decompositions carry 22 bits / are exact, every piece product is
exact in fp32, and a long dot product computed that way is as close to the float64 result as a plain fp32 dot product is.  The GPU
tests (test_split_gpu.py) check that the kernels implement exactly this."""
import torch


def _f16_split(v):
    hi = v.half().float()
    lo = (v - hi).half().float()
    return hi, lo


def _bf16_split(v):
    p0 = v.bfloat16().float()
    r = v - p0
    p1 = r.bfloat16().float()
    r = r - p1
    p2 = r.bfloat16().float()
    return p0, p1, p2


def test_two_fp16_pieces_carry_22_bits():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(1 << 20, generator=g) * torch.logspace(-4, 4, 1 << 20)          # eight decades inside fp16's range
    hi, lo = _f16_split(v)
    err = (v.double() - hi.double() - lo.double()).abs()
    assert (err <= v.double().abs() * 2.0 ** -22 + 2.0 ** -25).all()               # relative 2^-22 until lo reaches fp16's subnormal quantum
    assert (hi.abs() <= 65504).all()
    # each piece product is exact in fp32: an 11-bit by 11-bit significand product has 22 bits
    a, b = hi[:4096], lo[4096:8192]
    assert torch.equal((a * b).double(), a.double() * b.double())


def test_three_bf16_pieces_are_exact_over_fp32s_range():
    g = torch.Generator().manual_seed(1)
    v = torch.randn(1 << 20, generator=g) * torch.logspace(-25, 30, 1 << 20)         # 55 decades: gradients of any magnitude
    p0, p1, p2 = _bf16_split(v)
    # exactly: 8 + 8 + 8 significand bits — as long as the third piece (2^-16 of v) is a normal number, i.e. |v| >= 2^-110 = 8e-34
    # (below that it lands in the subnormals and the sum is off by <= 2^-149; three of a million values at 1e-35 showed it)
    assert torch.equal(p0.double() + p1.double() + p2.double(), v.double())
    # the three dropped piece products are <= 2^-24 of the full product
    a, b = v[: 1 << 19], v[1 << 19:]
    a0, a1, a2 = _bf16_split(a)
    b0, b1, b2 = _bf16_split(b)
    kept = (a0.double() * b0.double() + a0.double() * b1.double() + a1.double() * b0.double() +
            a0.double() * b2.double() + a1.double() * b1.double() + a2.double() * b0.double())
    full = a.double() * b.double()
    assert ((kept - full).abs() <= full.abs() * 2.0 ** -22).all()
    assert ((kept - full).abs() / full.abs().clamp_min(1e-300)).median().item() < 2.0 ** -25


def _errs(x, w):
    """max |err| / max |exact| of x @ w for: plain fp32, the fp16 split (3 products), the bf16 split (6 products)."""
    exact = x.double() @ w.double()
    scale = exact.abs().max().item()
    e32 = ((x @ w).double() - exact).abs().max().item() / scale
    xh, xl = _f16_split(x)
    wh, wl = _f16_split(w)
    y3 = (xl @ wh) + (xh @ wl) + (xh @ wh)                                           # fp32 accumulation of exact piece products
    e3 = (y3.double() - exact).abs().max().item() / scale
    x0, x1, x2 = _bf16_split(x)
    w0, w1, w2 = _bf16_split(w)
    y6 = (x0 @ w2) + (x2 @ w0) + (x1 @ w1) + (x0 @ w1) + (x1 @ w0) + (x0 @ w0)
    e6 = (y6.double() - exact).abs().max().item() / scale
    return e32, e3, e6


def test_split_dot_products_are_as_close_to_float64_as_fp32_ones():
    """A 1408-term reduction (128 channels x 11 taps, the generator's widest) and a 5120-term one (1024 x 5, the discriminators')."""
    torch.set_num_threads(4)
    for K, N, M, seed in ((1408, 256, 512, 2), (5120, 128, 256, 3)):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(K, N, generator=g) / K ** 0.5
        e32, e3, e6 = _errs(x, w)
        print(f"K={K}: fp32 {e32:.2e}, fp16 split x3 {e3:.2e}, bf16 split x6 {e6:.2e} (of max |exact|)")
        assert e3 <= 4 * e32 + 2e-7 and e6 <= 2 * e32 + 1e-7, (e32, e3, e6)
        assert e3 < 2e-6 and e6 < 2e-6


def test_fp16_pieces_lose_small_gradients_and_bf16_pieces_do_not():
    """Why training uses bf16 pieces: operands at 1e-7 (a late-layer gradient) sit in fp16's subnormals."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(256, 960, generator=g) * 1e-7
    w = torch.randn(960, 128, generator=g) / 960 ** 0.5
    e32, e3, e6 = _errs(x, w)
    assert e6 <= 2 * e32 + 1e-7
    assert e3 > 1e-3                                                                 # two or three significant bits left
