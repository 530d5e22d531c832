"""rpt_amd/csrc/sort_scan.h: the photon-map build's device-wide stable radix sort (63-bit Morton keys + the photon index) and its
two-array exclusive prefix sum, against numpy.  Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from rpt_amd import _lib

pytestmark = pytest.mark.gpu


def _sort(keys):
    n = len(keys)
    out_k = np.empty(n, dtype=np.uint64)
    out_o = np.empty(n, dtype=np.uint32)
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    _lib.check(_lib.load().rpt_debug_radix_sort(n, k.ctypes.data_as(C.c_void_p), out_k.ctypes.data_as(C.c_void_p), out_o.ctypes.data_as(C.c_void_p)))
    return out_k, out_o


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097, 100_003, 2_000_000])
def test_radix_sort_is_the_stable_sort(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    got_k, got_o = _sort(keys)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_o, order.astype(np.uint32)) and np.array_equal(got_k, keys[order])


@pytest.mark.parametrize("case", ["all equal", "two values", "few distinct", "sorted", "reversed", "one digit differs", "high bits only"])
def test_radix_sort_keeps_the_input_order_of_equal_keys(case):
    rng = np.random.default_rng(7)
    n = 70_001
    if case == "all equal":
        keys = np.full(n, 0x0123456789ABCDEF, dtype=np.uint64)
    elif case == "two values":
        keys = rng.integers(0, 2, size=n, dtype=np.uint64) * np.uint64(0x7FFFFFFFFFFFFFFF)
    elif case == "few distinct":
        keys = rng.choice(rng.integers(0, 1 << 63, size=17, dtype=np.uint64), size=n)
    elif case == "sorted":
        keys = np.sort(rng.integers(0, 1 << 63, size=n, dtype=np.uint64))
    elif case == "reversed":
        keys = np.sort(rng.integers(0, 1 << 63, size=n, dtype=np.uint64))[::-1].copy()
    elif case == "one digit differs":
        keys = (rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(24)) | np.uint64(0x00AA00BB00000011)
    else:
        keys = rng.integers(0, 128, size=n, dtype=np.uint64) << np.uint64(56)
    got_k, got_o = _sort(keys)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_o, order.astype(np.uint32)) and np.array_equal(got_k, keys[order])


@pytest.mark.parametrize("n", [0, 1, 255, 256, 2047, 2048, 2049, 524_288, 1_000_003])
def test_exclusive_scan_of_two_arrays(n):
    rng = np.random.default_rng(n + 1)
    a = rng.integers(0, 9, size=n, dtype=np.uint32)
    b = rng.integers(0, 3, size=n, dtype=np.uint32)
    oa, ob = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
    tot = (C.c_uint64 * 2)()
    _lib.check(_lib.load().rpt_debug_exclusive_scan2(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                     oa.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p), tot))
    ea = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]) if n else np.zeros(0)
    eb = np.concatenate([[0], np.cumsum(b, dtype=np.uint64)[:-1]]) if n else np.zeros(0)
    assert np.array_equal(oa, ea.astype(np.uint32)) and np.array_equal(ob, eb.astype(np.uint32))
    assert int(tot[0]) == int(a.sum(dtype=np.uint64)) and int(tot[1]) == int(b.sum(dtype=np.uint64))


def test_scan_totals_do_not_wrap_at_32_bits():
    n = 3_000_000
    a = np.full(n, 2000, dtype=np.uint32)          # sum = 6e9 > 2^32
    b = np.ones(n, dtype=np.uint32)
    oa, ob = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
    tot = (C.c_uint64 * 2)()
    _lib.check(_lib.load().rpt_debug_exclusive_scan2(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                     oa.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p), tot))
    assert int(tot[0]) == 2000 * n and int(tot[1]) == n and np.array_equal(ob, np.arange(n, dtype=np.uint32))
