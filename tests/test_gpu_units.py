"""Device-side arithmetic spec vs the oracle's, function by function (pytest -m gpu).
All comparisons are on raw bits."""
import ctypes as C

import numpy as np
import pytest

from skele_raytracer_amd import binding

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(1234)


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def test_philox(oracle):
    L = oracle.lib()
    inp = RNG.integers(0, 2**32, (4096, 6), dtype=np.uint64).astype(np.uint32)
    inp[0] = 0
    inp[1] = 0xFFFFFFFF
    inp[2] = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0]
    out = binding.debug_eval(0, inp, 4)   # the product's counter RNG: Philox4x32-7
    out10 = binding.debug_eval(8, inp, 4)  # the same round function at Random123's default of 10 rounds
    # Random123 known-answer vectors (kat_vectors: philox4x32 7 / philox4x32 10)
    assert out[0].tolist() == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert out[1].tolist() == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert out[2].tolist() == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    assert out10[0].tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert out10[1].tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert out10[2].tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    o = (C.c_uint32 * 4)()
    for i in range(0, 4096, 37):
        c = (C.c_uint32 * 4)(*inp[i, :4].tolist())
        k = (C.c_uint32 * 2)(*inp[i, 4:].tolist())
        L.sko_philox4x32_spec(c, k, o)
        assert list(o) == out[i].tolist()
        L.sko_philox4x32_10(c, k, o)
        assert list(o) == out10[i].tolist()


def test_sincos(oracle):
    L = oracle.lib()
    r2 = np.concatenate([RNG.integers(0, 2**31, 60000).astype(np.float32) / np.float32(2147483648.0),
                         f32([0.0, 1.0, 0.25, 0.5, 0.75, 1e-9, 0.125])])
    phi = (2.0 * np.pi * r2.astype(np.float64)).astype(np.float32)
    out = binding.debug_eval(1, phi.reshape(-1, 1), 2)
    s, c = C.c_float(), C.c_float()
    want = np.zeros((len(phi), 2), np.float32)
    for i, p in enumerate(phi):
        L.sko_sincos_shared(float(p), C.byref(s), C.byref(c))
        want[i] = (s.value, c.value)
    assert np.array_equal(out, want.view(np.uint32))


def test_powf(oracle):
    L = oracle.lib()
    x = np.concatenate([RNG.random(40000).astype(np.float32), f32([0, 1, 0.5, 1e-30, 0.99999994, 1.0000001, 2.0])])
    p = np.concatenate([RNG.choice(f32([16, 32, 2, 100, 20, 30, 0.5, 7.3, 1, 0]), 40000), f32([16, 16, 0, 32, 100, 32, 200])])
    out = binding.debug_eval(2, np.stack([x, p], 1), 1)
    want = f32([L.sko_powf_shared(float(a), float(b)) for a, b in zip(x, p)])
    assert np.array_equal(out[:, 0], want.view(np.uint32))


def test_smallest_root_fp64_path(oracle):
    """utils.h:87-110: the double-precision sqrt and divide must be correctly rounded on the device."""
    L = oracle.lib()
    n = 50000
    a = (RNG.random(n) * 2 + 0.01).astype(np.float32)
    b = ((RNG.random(n) - 0.7) * 200).astype(np.float32)
    c = ((RNG.random(n) - 0.3) * 3000).astype(np.float32)
    a[:5] = [1, 1, 1, 0, 1]
    b[:5] = [-2, 0, 5, -1, -4]
    c[:5] = [1, -1, 1, 1, 4]
    out = binding.debug_eval(3, np.stack([a, b, c], 1), 1)
    want = f32([L.sko_smallest_root(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)])
    assert np.array_equal(out[:, 0], want.view(np.uint32))


def test_triangle_predicate(oracle):
    L = oracle.lib()
    L.sko_triangle_test.argtypes = [C.POINTER(C.c_float)] * 5 + [C.POINTER(C.c_float)]
    L.sko_triangle_test.restype = C.c_int
    n = 20000
    rec = ((RNG.random((n, 15)) - 0.5) * 4).astype(np.float32)
    rec[:, 3:6] *= 0.5
    out = binding.debug_eval(4, rec, 2)
    t = C.c_float()
    hits = 0
    for i in range(0, n, 7):
        arrs = [(C.c_float * 3)(*rec[i, k:k + 3].tolist()) for k in (0, 3, 6, 9, 12)]
        h = L.sko_triangle_test(*arrs, C.byref(t))
        assert out[i, 0] == h
        if h:
            hits += 1
            assert out[i, 1] == np.float32(t.value).view(np.uint32)
    assert hits > 50


def test_quantise_and_basis(oracle):
    L = oracle.lib()
    L.sko_basis.argtypes = [C.POINTER(C.c_float)] * 3
    v = np.concatenate([RNG.random(5000).astype(np.float32) * 1.2, f32([0, 1, 0.999999, 1.5, np.nan, np.inf, 0.5, 254.999 / 255, 1 / 255])])
    out = binding.debug_eval(5, v.reshape(-1, 1), 1)
    assert out[:, 0].tolist() == [L.sko_quantise(float(x)) for x in v]
    nrm = RNG.normal(size=(4000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    nrm[0] = [0, 1, 0]
    nrm[1] = [1, 0, 0]
    nrm[2] = [0, 0, 1]
    ob = binding.debug_eval(6, nrm, 6)
    nt, nb = (C.c_float * 3)(), (C.c_float * 3)()
    for i in range(0, 4000, 5):
        L.sko_basis((C.c_float * 3)(*nrm[i].tolist()), nt, nb)
        assert np.array_equal(ob[i], f32(list(nt) + list(nb)).view(np.uint32))


def test_float_sqrt_and_divide_are_correctly_rounded():
    a = np.concatenate([RNG.random(100000).astype(np.float32) * 1000, f32([0, 1, 2, 3, 1e-40, 1e38, 0.1])])
    b = np.concatenate([(RNG.random(100000).astype(np.float32) - 0.5) * 50, f32([1, 3, 7, 0.1, 3, 1e-3, 3])])
    out = binding.debug_eval(7, np.stack([a, b], 1), 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        want_s = np.sqrt(a.astype(np.float64)).astype(np.float32)   # sqrt of a float in double, rounded once == correctly rounded
        want_d = (a.astype(np.float64) / b.astype(np.float64)).astype(np.float32)  # double quotient of floats rounds correctly (53 >= 2*24+2)
    assert np.array_equal(out[:, 0], want_s.view(np.uint32))
    assert np.array_equal(out[:, 1], want_d.view(np.uint32))


def test_short_exact_forms_equal_the_correctly_rounded_ones_on_every_float():
    """device_math.h sk_sqrtf / sk_rcpf (one Newton step on v_rsq / v_rcp for operands in [2^-100, 2^101), the compiler's expansion
    elsewhere) and div_const (x / pi, x / pdf as fma(x, zh, x * zl) for x == 0 or |x| >= 2^-100): every one of the 2^32 binary32
    operands, compared on the device with the expansions that test_float_sqrt_and_divide_are_correctly_rounded pins to IEEE."""
    hi = np.arange(65536, dtype=np.uint32).reshape(-1, 1)
    out = binding.debug_eval(9, hi, 4)
    assert out.sum(axis=0).tolist() == [0, 0, 0, 0], out.sum(axis=0)
