"""The arithmetic claim behind k_gru_gs / k_mfma_ls, checked in numpy (CPU): an fp32 operand is the EXACT sum of three bf16
terms (round to nearest even, remainders exact), a bf16 x bf16 product is exact in fp32, and of the nine term products of
w = w0 + w1 + w2 and h = h0 + h1 + h2 the three the kernels drop (w1 h2, w2 h1, w2 h2) are below fp32's own rounding of the
product. The device splits h with v_cvt_pk_bf16_f32 (round to nearest even), the packer splits the weights with
aidax::split_bf16x3 (checked term by term against the fp32 record under ASan, tests/asan_harness.cpp) — this file pins the
mathematics both rely on. Reference semantics it stands in for: RTNeural's fp32 dot products (rt-neural-generic.cpp:148-240)."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """x rounded to bf16 (round to nearest even), returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    r = np.where((r & 0x7F800000) == 0x7F800000, u & 0xFFFF0000, r)      # never round a finite value up to inf (as split_bf16x3 does)
    return r.astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    t0 = bf16_rne(x)
    r1 = x - t0                       # exact in fp32
    t1 = bf16_rne(r1)
    t2 = r1 - t1                      # exact, and itself a bf16
    return t0, t1, t2


def test_three_bf16_terms_are_the_fp32_value_exactly():
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.standard_normal(200000), rs.uniform(-1, 1, 200000) * 10.0 ** rs.uniform(-20, 30, 200000),      # (below ~1e-30 the third term would be an fp32 denormal)
                        [0.0, 1.0, -1.0, 1.1754944e-38, 3.4e38, 0.99999994, 1.0000001]]).astype(np.float32)
    t0, t1, t2 = split3(x)
    assert np.array_equal((t2.astype(np.float64) + t1) + t0, x.astype(np.float64))          # exact, not approximately
    assert np.array_equal(bf16_rne(t2), t2)                                                 # the third term IS a bf16
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(t1) <= ax * 2.0 ** -8 + 1e-45) and np.all(np.abs(t2) <= ax * 2.0 ** -16 + 1e-45)


def test_term_products_are_exact_in_fp32_and_the_dropped_ones_are_below_fp32_rounding():
    rs = np.random.RandomState(2)
    w = (rs.standard_normal(100000) * 0.3).astype(np.float32)
    h = rs.uniform(-1, 1, 100000).astype(np.float32)
    ws, hs = split3(w), split3(h)
    exact = w.astype(np.float64) * h.astype(np.float64)
    nine = np.zeros_like(exact)
    for a in ws:
        for b in hs:
            p32 = a * b                                                                     # fp32 multiply of two bf16 values
            assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))   # ... is exact (8 + 8 significant bits)
            nine += p32.astype(np.float64)
    assert np.array_equal(nine, exact)                                                      # all nine: the product to the last bit
    dropped = sum(ws[i].astype(np.float64) * hs[j] for i, j in ((1, 2), (2, 1), (2, 2)))
    assert np.all(np.abs(dropped) <= np.abs(exact) * 2.0 ** -23 + 1e-300)                   # six products: within one fp32 ulp of the product
    # and a whole dot product of cfg5's length (192 terms), six products accumulated in fp32 like the MFMA does, is as close to
    # the fp64 result as the plain fp32 dot product is
    W = (rs.standard_normal((2000, 192)) * 0.2).astype(np.float32)
    H = rs.uniform(-1, 1, (2000, 192)).astype(np.float32)
    ref = (W.astype(np.float64) * H).sum(1)
    plain = np.zeros(2000, np.float32)
    for k in range(192):
        plain = plain + W[:, k] * H[:, k]
    Ws, Hs = split3(W), split3(H)
    six = np.zeros(2000, np.float32)
    for i, j in ((0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)):
        six = six + (Ws[i].astype(np.float64) * Hs[j]).sum(1).astype(np.float32)            # one MFMA group: exact products, one fp32 rounding
    e_plain, e_six = np.abs(plain - ref).max(), np.abs(six - ref).max()
    assert e_six <= 2.0 * e_plain + 1e-7, (e_six, e_plain)
