"""The product's own arithmetic choices (what replaces rand() and libm's cosf/sinf), checked on the CPU twin that the device
is compared with bit for bit (tests/test_gpu_units.py): Random123's known-answer vectors for Philox4x32 at 7 rounds (what the
draws use) and at 10 (the round function at Random123's default), and the binary32 sincos against correctly rounded values."""
import ctypes as C

import numpy as np


def _philox(L, ctr, key, rounds):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    L.sko_philox4x32_r(c, k, rounds, o)
    return list(o)


def test_philox_known_answers(oracle):
    L = oracle.lib()
    pi = ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])
    ones = ([0xffffffff] * 4, [0xffffffff] * 2)
    zero = ([0] * 4, [0] * 2)
    # Random123 kat_vectors, "philox4x32 7" and "philox4x32 10"
    assert _philox(L, *zero, 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert _philox(L, *ones, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert _philox(L, *pi, 7) == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    assert _philox(L, *zero, 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox(L, *ones, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox(L, *pi, 10) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    o = (C.c_uint32 * 4)()
    L.sko_philox4x32_spec((C.c_uint32 * 4)(*pi[0]), (C.c_uint32 * 2)(*pi[1]), o)
    assert list(o) == _philox(L, *pi, 7)  # the draws use 7 rounds


def _ulps(got, want64):
    w = want64.astype(np.float32)
    ulp = np.spacing(np.abs(w)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 2.0 ** -149)


def test_sincos_shared_accuracy(oracle):
    """Every 1021st float of [0, 2 pi] plus the neighbourhoods of the quadrant boundaries: the binary32 recipe stays within
    1.5 ulp of the correctly rounded sine and cosine (tools/sincos_exhaustive.c measures 1.43 ulp over ALL 1 086 918 620 floats:
    profiles/r03_sincos_exhaustive.txt), and phi = 2 pi r2 of actual draws does too."""
    L = oracle.lib()
    top = int(np.float32(6.2831855).view(np.uint32))
    bits = np.arange(0, top + 1, 1021, dtype=np.uint32)
    edges = np.concatenate([np.float32(k * np.pi / 2).view(np.uint32) + np.arange(-300, 300, dtype=np.int64) for k in (1, 2, 3, 4)])
    edges = edges[(edges >= 0) & (edges <= top)].astype(np.uint32)
    rng = np.random.default_rng(3)
    r2 = rng.integers(0, 2 ** 31, 20000).astype(np.float32) / np.float32(2147483648.0)
    draws = (2.0 * np.pi * r2.astype(np.float64)).astype(np.float32)
    phi = np.concatenate([bits.view(np.float32), edges.view(np.float32), draws])
    s, c = C.c_float(), C.c_float()
    got = np.zeros((len(phi), 2), np.float32)
    for i, p in enumerate(phi):
        L.sko_sincos_shared(float(p), C.byref(s), C.byref(c))
        got[i] = (s.value, c.value)
    es, ec = _ulps(got[:, 0], np.sin(phi.astype(np.float64))), _ulps(got[:, 1], np.cos(phi.astype(np.float64)))
    assert es.max() < 1.5 and ec.max() < 1.5, (es.max(), ec.max())
    assert (es > 1.0).mean() < 2e-3 and (ec > 1.0).mean() < 2e-3
    # sin^2 + cos^2 and the quadrant logic: exact symmetries of the recipe
    L.sko_sincos_shared(0.0, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (0.0, 1.0)
