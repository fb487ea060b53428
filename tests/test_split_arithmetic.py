"""The arithmetic claim behind gemm_mode 2 (ppo_grad_split_kernel), checked in numpy with an emulated bf16 (no GPU):
a float32 x is carried as three bf16 planes x = h + m + l -- exactly -- and a product as six plane products accumulated in
float32.  The GPU-side measurement of the same claim is tests/test_gpu_parity.py (against a float64 gradient) and
profiles/r03_split_bf16_probe.txt."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2), r1, r2


def test_three_bf16_planes_carry_a_float32_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200_000).astype(np.float32),
                        (rng.standard_normal(50_000) * 1e-6).astype(np.float32),
                        (rng.standard_normal(50_000) * 1e6).astype(np.float32),
                        np.tanh(rng.standard_normal(100_000)).astype(np.float32),
                        np.float32([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e-30, 65504.0, 2.0 ** -100])])
    h, m, l, r1, r2 = split3(x)
    # the residues are exact in float32 (that is why two subtractions suffice) ...
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    # ... and the third plane takes the last residue without rounding: h + m + l == x, bit for bit
    assert np.array_equal(l, r2)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # each plane is a bf16: 8 significant bits, low 16 bits of the float32 pattern clear
    for p in (h, m, l):
        assert not (p.view(np.uint32) & 0xFFFF).any()
    # magnitudes fall off by 2^-8 per plane (round-to-nearest residues)
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def _matmul_f32_chain(a, b):
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):        # k-ordered fmaf chain (float64 product of two float32 is exact; one rounding per step)
        acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)
    return acc


def _matmul_split(a, b, terms):
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    pa, pb = (ah, am, al), (bh, bm, bl)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i, j in terms:                 # plane products are exact in float32 (8 x 8 significant bits); float32 accumulation per k
        for k in range(a.shape[1]):
            acc = (acc.astype(np.float64) + pa[i][:, k:k + 1].astype(np.float64) * pb[j][k:k + 1, :].astype(np.float64)).astype(np.float32)
    return acc


SIX = ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))      # the kernel's order: small terms first
NINE = ((2, 2), (1, 2), (2, 1)) + SIX


def test_six_term_product_is_float32_accurate_and_the_dropped_terms_do_not_matter():
    rng = np.random.default_rng(1)
    for scale_a, scale_b in ((1.0, 1.0), (1.0, 0.2)):
        a = (rng.standard_normal((48, 64)) * scale_a).astype(np.float32)
        b = (rng.standard_normal((64, 48)) * scale_b).astype(np.float32)
        if scale_b != 1.0:
            a = np.tanh(a).astype(np.float32)                  # activation x weight
        ref = a.astype(np.float64) @ b.astype(np.float64)
        bound = (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64))      # sum_k |a||b| per output element
        e32 = np.abs(_matmul_f32_chain(a, b) - ref)
        e6 = np.abs(_matmul_split(a, b, SIX) - ref)
        e9 = np.abs(_matmul_split(a, b, NINE) - ref)
        e3 = np.abs(_matmul_split(a, b, ((0, 1), (1, 0), (0, 0))) - ref)
        # dropped cross terms: ml + lm + ll <= (2 * 2^-24 + 2^-32) sum |a||b| -- far below the accumulation error of either path
        assert (np.abs(_matmul_split(a, b, NINE).astype(np.float64) - _matmul_split(a, b, SIX).astype(np.float64))
                <= 2.0 ** -22 * bound + 1e-30).all()
        # float32 level: within a small multiple of the exact-f32 chain's own error (here the emulation rounds after EVERY plane
        # product, which is harsher than the matrix pipe, which adds the 32 products of an instruction first)
        assert e6.max() <= 8 * e32.max() and np.sqrt((e6 ** 2).mean()) <= 8 * np.sqrt((e32 ** 2).mean())
        assert e9.max() <= 8 * e32.max()
        # ... while three terms (16 significant bits per operand) are an order of magnitude off: all three planes are needed
        assert e3.max() >= 10 * e32.max()
