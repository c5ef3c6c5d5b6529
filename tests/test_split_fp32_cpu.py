"""The arithmetic of the split-fp32 convolution products (u2pl_amd/csrc/conv.hip, BF == 3 / SP == 3), restated in numpy:
the three-piece bf16 split is EXACT, and the six retained piece products reproduce an fp32 dot product to fp32 accuracy.
(The GPU kernels are held to the same statement against float64 convolutions in tests/test_gpu_conv_stack.py.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 -- what v_cvt_pk_bf16_f32 does for finite values"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


BF16_MAX = np.float32(3.3895313892515355e38)      # 0x7F7F0000


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    # first piece: the conversion's overflow is taken out (conv_geom.h: clamp_bf16 / pack2_bf16_first) -- a finite |x| above
    # the largest bf16 would round to Inf and make the residual NaN
    x0 = bf16_rne(np.clip(x, -BF16_MAX, BF16_MAX))
    r1 = (x - x0).astype(np.float32)          # exact in fp32
    x1 = bf16_rne(r1)
    r2 = (r1 - x1).astype(np.float32)         # exact in fp32
    x2 = bf16_rne(r2)
    return x0, x1, x2, r1, r2


def test_three_piece_split_is_exact_and_each_piece_is_a_bfloat16():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * s for s in (1e-6, 1e-3, 1.0, 37.0, 1e5)]
                       + [np.float32([0.0, -0.0, 1.0, -1.0, 3.0000002, 1e-30, 6.5e4, 2.0 ** -100])])
    x0, x1, x2, r1, r2 = split3(x)
    for p in (x0, x1, x2):
        assert not (p.view(np.uint32) & 0xFFFF).any()                        # 8 significand bits each
    # the two residuals are exactly representable (computed here in float64 for the check)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    total = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    err = np.abs(total - x.astype(np.float64))
    assert np.all(err <= np.abs(x.astype(np.float64)) * 2.0 ** -24)           # 3 x 8 bits cover the 24-bit significand
    assert np.mean(err == 0) > 0.99                                           # ... and almost always exactly


def test_six_piece_products_give_an_fp32_class_dot_product():
    """sum_k a_k b_k over K = 2304 (a 3x3 conv over 256 channels) per output: the six piece products of weight >= 2^-16,
    accumulated like the matrix core does (fp32 accumulator over k-blocks of 16), against float64 -- next to a plain fp32
    FMA chain over the same data.  The split form must not be less accurate than the fp32 chain."""
    rng = np.random.default_rng(1)
    M, K = 512, 2304
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    ref = np.einsum("mk,mk->m", a.astype(np.float64), b.astype(np.float64))
    a0, a1, a2, _, _ = split3(a)
    b0, b1, b2, _, _ = split3(b)
    acc = np.zeros(M, dtype=np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        for pa, pb in ((a2, b0), (a1, b1), (a0, b2), (a1, b0), (a0, b1), (a0, b0)):       # smallest weights first
            # (a block of 16 exact bf16 x bf16 products summed at >= fp32 precision, then one fp32 accumulate)
            acc = (acc.astype(np.float64) + np.einsum("mk,mk->m", pa[:, s].astype(np.float64), pb[:, s].astype(np.float64))).astype(np.float32)
    chain = np.zeros(M, dtype=np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)   # fmaf chain
    scale = np.abs(ref).max()
    e_split, e_chain = np.abs(acc - ref).max() / scale, np.abs(chain - ref).max() / scale
    assert e_split <= 1.25 * e_chain + 1e-8, (e_split, e_chain)
    assert e_split < 2e-6
    # dropping one of the 2^-16 terms is visible: the six-term choice is the minimal one
    acc5 = np.zeros(M, dtype=np.float64)
    for pa, pb in ((a2, b0), (a0, b2), (a1, b0), (a0, b1), (a0, b0)):
        acc5 += np.einsum("mk,mk->m", pa.astype(np.float64), pb.astype(np.float64))
    assert np.abs(acc5 - ref).max() / scale > 4 * e_split


def test_split_is_exact_for_finite_values_above_the_largest_bfloat16():
    """3.3895e38 < |x| <= 3.4028e38: plain bf16(x) is +-Inf for the upper half of that range; with the clamp the first piece
    is +-bf16max and the other two carry the rest exactly"""
    fmax = np.finfo(np.float32).max
    x = np.float32([fmax, -fmax, np.nextafter(BF16_MAX, np.float32(np.inf)), 3.395e38, -3.399e38, 3.40e38, BF16_MAX, -BF16_MAX])
    assert np.isinf(bf16_rne(x[:2])).all()                                    # what the unguarded conversion does
    x0, x1, x2, r1, r2 = split3(x)
    assert np.isfinite(x0).all() and np.isfinite(x1).all() and np.isfinite(x2).all()
    assert np.array_equal(np.abs(x0), np.full(len(x), BF16_MAX))
    total = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))


# ---- split-fp16 (round 6; u2pl_amd/csrc/conv_geom.h): two fp16 pieces of the power-of-two-scaled operand, three piece products ----
def split2_exp(amax):
    """the kernels' scale exponent: amax * 2^e lies in [2^14, 2^15) (split2_exp_bits: from the exponent field of max |x|)"""
    ex = int((np.float32(amax).view(np.uint32) >> 23) & 0xFF)
    return min(14 - (ex - 127), 126)


def split2(x, e):
    xs = (np.asarray(x, dtype=np.float32) * np.float32(2.0) ** e).astype(np.float32)      # exact (power of two)
    with np.errstate(over="ignore"):
        h0 = xs.astype(np.float16)                                                         # round to nearest even, subnormals kept
    r = (xs - h0.astype(np.float32)).astype(np.float32)                                    # exact in fp32
    h1 = r.astype(np.float16)
    return xs, h0, h1


def test_two_fp16_pieces_carry_an_fp32_value_to_one_fp32_ulp():
    """h0 + h1 = x s to within 2^-23 |x s| (one fp32 ulp: TWICE the rounding fp32 itself commits; rms 2^-24.4 against fp32's 2^-25.2,
    unbiased): h0 keeps 11 bits, the residual is at most 2^-11 |x s| and h1 keeps 11 bits of IT"""
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * s for s in (1e-3, 1.0, 37.0)])
    e = split2_exp(np.abs(x).max())
    xs, h0, h1 = split2(x, e)
    assert 2.0 ** 14 <= np.abs(xs).max() < 2.0 ** 15 and np.isfinite(h0.astype(np.float32)).all()
    err = np.abs(xs.astype(np.float64) - h0.astype(np.float64) - h1.astype(np.float64))
    big = np.abs(xs) >= 2.0 ** -2                      # >= 2^-17 of the maximum: a subnormal second piece is still good to 2^-25 absolute
    rel = err[big] / np.abs(xs[big].astype(np.float64))
    assert rel.max() <= 2.0 ** -23 and np.sqrt((rel ** 2).mean()) <= 2.0 ** -24
    # below that: an ABSOLUTE error of half the fp16 subnormal spacing, 2^-25 (= 2^-40 of the maximum)
    assert np.all(err <= np.maximum(np.abs(xs.astype(np.float64)) * 2.0 ** -23, 2.0 ** -25))


def test_three_fp16_piece_products_give_an_fp32_class_dot_product():
    """the GEMMs' arithmetic restated: a1 b0 + a0 b1 + a0 b0 per 16-deep block, fp32 accumulate, scaled back by 2^-(ea + eb) --
    against float64, next to the six-product bf16 form and a plain fp32 chain on the same data"""
    rng = np.random.default_rng(3)
    M, K = 512, 2304
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)                       # post-ReLU activations
    b = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    ref = np.einsum("mk,mk->m", a.astype(np.float64), b.astype(np.float64))
    ea, eb = split2_exp(np.abs(a).max()), split2_exp(np.abs(b).max())
    _, a0, a1 = split2(a, ea)
    _, b0, b1 = split2(b, eb)
    acc = np.zeros(M, dtype=np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        for pa, pb in ((a1, b0), (a0, b1), (a0, b0)):
            acc = (acc + np.einsum("mk,mk->m", pa[:, s].astype(np.float64), pb[:, s].astype(np.float64)).astype(np.float32)).astype(np.float32)
    got = np.ldexp(acc.astype(np.float64), -(ea + eb))
    chain = np.zeros(M, dtype=np.float32)
    for k in range(K):
        chain = (chain + a[:, k] * b[:, k]).astype(np.float32)
    scale = np.abs(ref).max()
    e3 = np.abs(got - ref).max() / scale
    ec = np.abs(chain.astype(np.float64) - ref).max() / scale
    assert e3 <= ec and e3 < 1e-6, (e3, ec)


def test_scale_exponent_edge_cases():
    assert split2_exp(1.0) == 14 and split2_exp(1.9999999) == 14 and split2_exp(2.0) == 13
    assert split2_exp(3.0e38) == 14 - 127                      # the largest finite fp32 still lands below 2^15
    assert np.float32(3.4e38) * np.float32(2.0) ** split2_exp(3.4e38) < 2.0 ** 15
    assert split2_exp(0.0) == 126 and split2_exp(1e-45) == 126 # zero / subnormal maxima: the largest normal scale
    assert split2_exp(np.float32(np.inf)) == 14 - 128          # Inf / NaN maxima: a normal scale, the pieces come out Inf / NaN
