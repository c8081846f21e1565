"""The building block of the exact kernels' replacement (DESIGN.md "Open ends"; profiles/r04/full_cert_experiment): the reference
aborts a pair when a row's maximum of fl(best + pen(k)) is below -600 (HapAligner.cpp:297-306), pen(k) = (double)((float)|k| * c).
x -> fl(x + p) is monotone, so the row survives iff some cell has best >= thr(k), thr(k) the SMALLEST double with
fl(thr + pen(k)) >= -600.  This test builds thr(k) the way the prototype's table does (bisection over bit patterns) and checks
the equivalence around every threshold."""
import numpy as np
import pytest

C_VALUES = [-1.0, -0.458675, -0.9, -4.6, -0.75]                 # LOG_DEL_TO_DEL of the parameter sets the tests and the fuzz use


def pen(k, c32):
    return np.float64(np.float32(k) * np.float32(c32))            # int * float -> float, then promoted (HapAligner.cpp:298)


def threshold(k, c32):
    """Bisection over the bit patterns of the (negative) doubles around -600 - p: the predicate is monotone.  Steps of one ulp
    from the guess do NOT get there: at k = 597, c = -1 the guess is -3 and the threshold 128 ulps of it further down (half an
    ulp of 600)."""
    p = pen(k, c32)
    x0 = np.float64(-600.0) - p
    if not x0 < -1e-6:
        return np.inf                                            # every cell is < 0: "never"
    ok, bad = np.float64(x0 + 1e-9), np.float64(x0 - 1e-9)       # ok passes, bad fails; both negative
    assert ok + p >= -600.0 and not (bad + p >= -600.0)
    lo, hi = int(ok.view(np.int64)), int(bad.view(np.int64))     # negative doubles: larger pattern = further down
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if np.int64(mid).view(np.float64) + p >= -600.0:
            lo = mid
        else:
            hi = mid
    return np.int64(lo).view(np.float64)


@pytest.mark.parametrize("c32", C_VALUES)
def test_threshold_is_the_exact_inverse_of_the_penalised_compare(c32):
    cabs = abs(np.float32(c32))
    k600 = int(np.float32(600.0) / cabs) + 2
    rng = np.random.default_rng(7)
    for k in range(0, min(k600, 1024)):
        p, thr = pen(k, c32), threshold(k, c32)
        if np.isinf(thr):
            assert not (np.float64(-1e-6) + p >= -600.0)         # nothing negative passes either
            continue
        assert thr + p >= -600.0 and not (np.nextafter(thr, -np.inf) + p >= -600.0)
        # the compare the reference makes and the compare against the table agree on every double around the threshold
        x = thr
        for _ in range(8):
            x = np.nextafter(x, -np.inf)
        for _ in range(17):
            assert (x + p >= -600.0) == (x >= thr)
            x = np.nextafter(x, np.inf)
        xs = thr + rng.normal(0.0, 1e-12, 32)
        assert np.array_equal(xs + p >= -600.0, xs >= thr)


def test_beyond_k600_no_negative_cell_passes():
    for c32 in C_VALUES:
        cabs = abs(np.float32(c32))
        k600 = int(np.float32(600.0) / cabs) + 2
        for k in (k600, k600 + 1, 2 * k600):
            assert not (np.float64(-0.0001) + pen(k, c32) >= -600.0)


@pytest.mark.parametrize("c32", C_VALUES)
def test_library_table_is_that_construction(c32):
    """ltr_debug_threshold_table = ltrp::build_threshold_table, the table the LUT exact kernels copy into LDS."""
    import ctypes as C
    from longtr_amd import _lib
    L = _lib.lib()
    L.ltr_debug_threshold_table.argtypes = [C.c_float, C.c_void_p, C.c_int64]
    L.ltr_debug_threshold_table.restype = C.c_int
    out = np.zeros(4096, dtype=np.float64)
    n = L.ltr_debug_threshold_table(C.c_float(c32), out.ctypes.data, out.size)
    assert n > 0 and n % 2 == 0
    half = n // 2
    cabs = abs(np.float32(c32))
    k600 = int(np.float32(600.0) / cabs) + 2
    for idx in range(n):
        k = abs(idx - half)
        want = threshold(k, c32) if (k < k600 and k <= 1023) else np.inf
        assert out[idx] == want or (np.isinf(out[idx]) and np.isinf(want)), (idx, k, out[idx], want)
    assert L.ltr_debug_threshold_table(C.c_float(c32), out.ctypes.data, 16) < 0      # too small a buffer
