"""The fp16 x 2 arithmetic of the 16x16 Winograd and pointwise kernels (include/p2l.h,
P2L_WFMT_BF16X3W / _PW; csrc/p2l_conv_k.h h2_scales), restated in numpy: what the scaling and the
two-piece split guarantee, independent of any kernel.  (CPU test: float16 conversions of numpy are
round-to-nearest-even like v_cvt_pk_f16_f32.)"""
import numpy as np
import pytest


def h2_scales(amax):
    """power of two that puts 4 * amax (headroom of the Winograd input transform) below 2^15, and
    its inverse -- the integer arithmetic of h2_scales() on the exponent field"""
    bits = np.float32(amax).view(np.uint32)
    E = int((bits >> 23) & 0xff)
    E = min(max(E, 40), 254)
    scale = np.uint32((266 - E) << 23).view(np.float32)
    inv = np.uint32((E - 12) << 23).view(np.float32)
    return scale, inv


def split2(x32):
    h = x32.astype(np.float16)
    m = (x32 - h.astype(np.float32)).astype(np.float16)
    return h, m


@pytest.mark.parametrize('mag', [1e-30, 1e-12, 3e-5, 1.0, 777.0, 1e6, 1e30])
def test_scale_is_an_exact_power_of_two_with_headroom(mag):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4096) * mag).astype(np.float32)
    amax = np.abs(x).max()
    s, inv = h2_scales(amax)
    assert np.float32(s) * np.float32(inv) == np.float32(1.0)                 # exact inverse
    assert np.log2(float(s)) == round(np.log2(float(s)))                      # a power of two
    if amax >= 2.0 ** -87:                                                    # (below: clamped, harmless)
        top = 4.0 * float(amax) * float(s)
        assert 2.0 ** 13 <= top < 2.0 ** 15                                   # transform-domain values < 2^15
    xs = x * s
    assert np.array_equal(xs * inv, x)                                        # scaling loses nothing
    assert np.isfinite(xs.astype(np.float16)).all()


def test_two_pieces_carry_an_fp32_value_to_one_ulp_relative_to_the_image_maximum():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(1 << 16) * np.exp(rng.uniform(-12, 0, 1 << 16))).astype(np.float32)
    s, inv = h2_scales(np.abs(x).max())
    xs = x * s
    h, m = split2(xs)
    back = h.astype(np.float64) + m.astype(np.float64)
    err = np.abs(back - xs.astype(np.float64))
    big = np.abs(xs) >= 0.25                       # second piece still a normal fp16 number
    assert (err[big] <= 2.0 ** -23 * np.abs(xs[big])).all()
    assert (err <= 2.0 ** -25).sum() + big.sum() >= err.size          # the rest: absolute 2^-25 (denormal step / 2)
    assert err.max() <= max(2.0 ** -25, 2.0 ** -23 * np.abs(xs).max())
    # relative to the image maximum (>= 2^12 after scaling) the small elements are accurate to 2^-37
    assert (err[~big] / np.abs(xs).max()).max() <= 2.0 ** -37


def test_three_products_are_fp32_grade():
    """h_a h_b + h_a m_b + m_a h_b (fp32 accumulate) against the exact product sum: the dropped m m
    term and the representation error stay at a few ulp of fp32 -- the level of an fp32 FMA chain"""
    rng = np.random.default_rng(2)
    K = 512
    a = rng.standard_normal((256, K)).astype(np.float32)
    b = (rng.standard_normal((K, 64)) / np.sqrt(K)).astype(np.float32)
    sa, ia = h2_scales(np.abs(a).max())
    sb, ib = h2_scales(np.abs(b).max())
    ha, ma = split2(a * sa)
    hb, mb = split2(b * sb)
    f = lambda t: t.astype(np.float32)
    d = lambda t: t.astype(np.float64)
    un = float(ia) * float(ib)
    exact = d(a) @ d(b)
    scale = np.abs(exact).max()
    # the arithmetic itself (exact accumulation of the three products): 9e-8 of the output range here
    acc64 = (d(ma) @ d(hb) + d(ha) @ d(mb) + d(ha) @ d(hb)) * un
    assert np.abs(acc64 - exact).max() / scale < 2e-7
    # with fp32 accumulation it sits in the noise of an fp32 matmul of the same data (5.6e-7 | 5.0e-7)
    acc = (f(ma) @ f(hb) + f(ha) @ f(mb) + f(ha) @ f(hb)) * np.float32(un)
    e_h2 = np.abs(acc - exact).max() / scale
    e_32 = np.abs(a @ b - exact).max() / scale
    assert e_h2 < 2 * e_32 + 1e-7


def test_results_do_not_depend_on_the_power_of_two():
    """a loose bound (the consumer of a tensor with a fused prologue scales by max|s| max|x| + max|t|)
    costs range, not precision: the same values come out for scales 2^k apart as long as the pieces
    stay normal fp16 numbers"""
    rng = np.random.default_rng(3)
    x = (1.0 + rng.random(4096)).astype(np.float32)                          # one binade: no denormal pieces
    outs = []
    for k in (0, 3, 6):
        s = np.float32(2.0 ** (10 - k))
        h, m = split2(x * s)
        outs.append((h.astype(np.float64) + m.astype(np.float64)) / float(s))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
