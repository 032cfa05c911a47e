"""Numerics of the two-plane fp16 split behind the matrix-core products (kaldi-lstm_amd/csrc/klstm_math.h f16_split2 / f16_split2_pair;
DESIGN.md 4b, 9 item 5), restated in numpy -- no GPU needed.  x = h1 + h2 / 2048 with h1 = fp16(x) (round to nearest even) and
h2 = fp16((x - h1) * 2048); the product kernels add a1 b1 + (a1 b2 + a2 b1) / 2048 in fp32 and drop a2 b2 / 2048^2."""
import numpy as np


def split2(x):
    x = np.asarray(x, np.float32)
    h1 = x.astype(np.float16)
    r = (x - h1.astype(np.float32)).astype(np.float32)            # exact in fp32
    h2 = (r * np.float32(2048.0)).astype(np.float16)
    return h1, h2


def test_split_reconstructs_to_22_bits_over_the_normal_fp16_range():
    rng = np.random.RandomState(0)
    for scale in (1e-4, 1e-2, 1.0, 50.0, 3e4):
        x = (rng.randn(200000) * scale).astype(np.float32)
        x = x[np.abs(x) >= 2.0 ** -14]                            # below: the absolute error bound applies (next test)
        x = x[np.abs(x) < 65504.0]
        h1, h2 = split2(x)
        assert np.all(np.isfinite(h1.astype(np.float32))) and np.all(np.isfinite(h2.astype(np.float32)))
        recon = h1.astype(np.float64) + h2.astype(np.float64) / 2048.0
        assert np.max(np.abs(recon - x.astype(np.float64)) / np.abs(x)) <= 2.0 ** -22


def test_split_of_tiny_values_has_a_tiny_absolute_error():
    x = (np.random.RandomState(1).randn(100000) * 1e-6).astype(np.float32)     # fp16 subnormals and below
    h1, h2 = split2(x)
    recon = h1.astype(np.float64) + h2.astype(np.float64) / 2048.0
    assert np.max(np.abs(recon - x.astype(np.float64))) <= 2.0 ** -35          # (2^-36 per plane rounding, klstm_math.h)


def test_the_2048_scale_is_what_keeps_small_values_accurate():
    """|x - h1| <= 2^-11 |x| falls into fp16's subnormal range for |x| < 2^-3: stored unscaled, the second plane would lose bits
    there (values around 2^-10: ~2^-14 relative); scaled by 2^11 it keeps the full 2^-22."""
    x = (np.random.RandomState(2).rand(100000).astype(np.float32) + 0.5) * np.float32(2.0 ** -10)
    h1, h2 = split2(x)
    recon = h1.astype(np.float64) + h2.astype(np.float64) / 2048.0
    unscaled = (x - h1.astype(np.float32)).astype(np.float16)
    recon_unscaled = h1.astype(np.float64) + unscaled.astype(np.float64)
    err = np.max(np.abs(recon - x) / np.abs(x)); err_unscaled = np.max(np.abs(recon_unscaled - x) / np.abs(x))
    assert err <= 2.0 ** -22 and err_unscaled >= 2.0 ** -16


def test_three_products_match_fp32_accuracy():
    """Dot products of 512 terms: the three-product sum (fp32 accumulation emulated in float64 here -- the partial products of two
    11-bit significands are exact in fp32) against float64, relative to sum |a_k b_k|: the dropped a2 b2 term is ~2^-22."""
    rng = np.random.RandomState(3)
    a = rng.randn(64, 512).astype(np.float32); b = (0.1 * rng.randn(512, 64)).astype(np.float32)
    a1, a2 = split2(a); b1, b2 = split2(b)
    f = lambda h: h.astype(np.float64)
    got = f(a1) @ f(b1) + (f(a1) @ f(b2) + f(a2) @ f(b1)) / 2048.0
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(got - ref) / scale) <= 2.0 ** -21
    # each partial product a1*b1 has at most 22 significant bits: exact in fp32
    p = (a1[:, :8].astype(np.float32)[:, :, None] * b1[:8, :].astype(np.float32)[None, :, :])
    assert np.array_equal(p.astype(np.float64), f(a1[:, :8])[:, :, None] * f(b1[:8, :])[None, :, :])


def test_values_beyond_the_fp16_range_overflow():
    """The documented limit (klstm.h "fp16_products"): 65520 and above round to infinity."""
    with np.errstate(over="ignore", invalid="ignore"):
        h1, _ = split2(np.array([65519.0, 65520.0, 1e5], np.float32))
    assert np.isfinite(h1[0]) and np.isinf(h1[1]) and np.isinf(h1[2])
