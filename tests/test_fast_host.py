"""The counter-based PLAN engine (``MSIM_RNG_FAST`` / ``--rng fast``) on the CPU tier.

It is NOT stream-compatible with the reference by design -- bit parity is impossible and is not claimed.  What is held here:

* the arithmetic of ``csrc/fast_math.h`` (compiled for the host) against the numpy restatement ``tests/fast_twin.py``,
  bit for bit: logarithm, square root, log-factorial differences, the hypergeometric sampler;
* the hypergeometric sampler against the exact law (``scipy.stats.hypergeom``) by chi-square;
* ``msim_dbg_fast_plan`` -- the library's sequential restatement of the whole engine over that arithmetic -- against the
  twin: every record, insert pool byte and the plan-was-empty flag, for SNP-only ranges, SV mixes, several ranges with
  their own settings (visit filter across range borders), ``sn_block`` above the sampling distance, dense hot spots
  (complement sampling), IV drops and DE / DU clamps at the contig end.

The GPU tier (``tests/test_gpu_fast_rng.py``) holds the kernels against the same twin and compares the mode's
distributions with the ORACLE's."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import fast_twin as ft
from mutation_simulator_amd import _ffi
from test_multimix_host import _engine, _range

C3_CHANCES = {1: 0.005, 2: 0.001, 3: 0.001, 4: 0.0005, 5: 0.0005}
C3_LENS = {2: (1, 50), 3: (1, 50), 4: (50, 500), 5: (50, 500)}


def _lib():
    lib = _ffi.load()
    lib.msim_dbg_fast_math.restype = C.c_int
    lib.msim_dbg_fast_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.msim_dbg_fast_hypergeom.restype = C.c_int
    lib.msim_dbg_fast_hypergeom.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint64, C.c_void_p]
    lib.msim_dbg_fast_plan.restype = C.c_int
    lib.msim_dbg_fast_plan.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_ffi.Range), C.c_int, C.c_uint64, C.c_uint32, C.c_void_p,
                                       C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int)]
    return lib


def _math(op, x, y=None):
    lib = _lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = C.c_void_p(y.ctypes.data)
    assert lib.msim_dbg_fast_math(op, C.c_void_p(x.ctypes.data), yp, len(x), C.c_void_p(out.ctypes.data)) == 0
    return out


def _same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))


def test_log_sqrt_log_factorial_bit_identical_and_accurate():
    from scipy.special import gammaln
    rng = np.random.RandomState(5)
    x = np.concatenate((np.exp(rng.uniform(-40, 700, 20000)), rng.uniform(0.5, 2.0, 20000), [1.0, 2.0, 0.5, 1.4142135623730951]))
    assert _same_bits(_math(0, x), ft.d_log(x))
    assert np.allclose(ft.d_log(x), np.log(x), rtol=4e-16, atol=2e-16)
    xs = np.exp(rng.uniform(-2, 60, 20000))
    assert _same_bits(_math(1, xs), ft.d_sqrt_up(xs))
    up = ft.d_sqrt_up(xs) / np.sqrt(xs)                            # an upper bound within 3e-4 (the sampler's hat only needs that)
    assert up.min() >= 1.0 - 1e-15 and up.max() < 1.0 + 3e-4
    n = np.concatenate((np.arange(0, 200), rng.randint(200, 2 ** 32, 5000))).astype(np.float64)
    assert _same_bits(_math(2, n), ft.log_factorial(n.astype(np.int64)))
    assert np.allclose(ft.log_factorial(n.astype(np.int64)), gammaln(n + 1.0), rtol=1e-14, atol=1e-12)
    t = np.concatenate((rng.uniform(-0.9, 5.0, 5000), rng.uniform(-1e-9, 1e-9, 5000), [0.0, 1e-300 * 0 + 2 ** -60]))
    assert _same_bits(_math(3, t), ft.d_log1p(t))
    assert np.allclose(ft.d_log1p(t), np.log1p(t), rtol=1e-15, atol=1e-300)
    a = np.concatenate((rng.randint(0, 200, 4000), rng.randint(200, 2 ** 32, 4000))).astype(np.int64)
    d = np.concatenate((rng.randint(-60, 60, 4000), rng.randint(-20000, 20000, 4000))).astype(np.int64)
    d = np.maximum(d, -a)
    got = _math(4, a.astype(np.float64), d.astype(np.float64))
    assert _same_bits(got, ft.log_factorial_diff(a, d))
    # accuracy where the plain difference is itself accurate (small arguments), and against exact sums of logs for small d
    small = a < 5000
    assert np.allclose(got[small], gammaln(a[small] + d[small] + 1.0) - gammaln(a[small] + 1.0), rtol=1e-12, atol=1e-10)
    big = (a > 10 ** 6) & (np.abs(d) <= 50)
    exact = np.array([np.sum(np.log(np.arange(ai + 1, ai + di + 1, dtype=np.float64))) if di >= 0 else
                      -np.sum(np.log(np.arange(ai + di + 1, ai + 1, dtype=np.float64))) for ai, di in zip(a[big], d[big])])
    assert np.allclose(got[big], exact, rtol=1e-13, atol=1e-12)


def _hyp(good, bad, sample, key, seq, node0, rng_id, n):
    out = np.zeros(n, dtype=np.uint64)
    assert _lib().msim_dbg_fast_hypergeom(good, bad, sample, key, seq, node0, rng_id, n, C.c_void_p(out.ctypes.data)) == 0
    return out.astype(np.int64)


HYP_CASES = [(100, 50, 7), (100, 50, 140), (5, 2 ** 31, 2 ** 30), (16384 * 700, 16384 * 300, 70_000),
             (2 ** 27, 2 ** 27, 2_400_000), (2 ** 27, 2 ** 27 + 12345, 2 ** 27), (65536, 65536 * 3000, 1_500_000),
             (1000, 1000, 1000), (12, 900, 400), (2 ** 31 - 5, 2 ** 31 - 9, 20_000_000), (40, 25, 33), (3, 4, 2)]


@pytest.mark.parametrize("good,bad,sample", HYP_CASES)
def test_hypergeometric_equals_twin_and_follows_the_exact_law(good, bad, sample):
    from scipy import stats
    n, key, seq = 20_000, 0x1234ABCD5678EF01, 9
    got = _hyp(good, bad, sample, key, seq, 77, 3, n)
    want = ft.hypergeometric(good, bad, sample, key, seq, 77 + np.arange(n), 3)
    assert np.array_equal(got, want)
    lo, hi = max(0, sample - bad), min(sample, good)
    assert got.min() >= lo and got.max() <= hi
    dist = stats.hypergeom(good + bad, good, sample)
    mean, sd = dist.mean(), dist.std()
    assert abs(got.mean() - mean) < 5 * sd / np.sqrt(n) + 1e-9
    # chi-square over bins of roughly equal mass
    qs = np.unique(dist.ppf(np.linspace(0, 1, 21)[1:-1]).astype(np.int64))
    edges = np.concatenate(([lo - 1], qs, [hi]))
    edges = np.unique(edges)
    obs = np.array([((got > a) & (got <= b)).sum() for a, b in zip(edges[:-1], edges[1:])])
    exp = n * np.diff(dist.cdf(edges))
    keep = exp > 5
    if keep.sum() >= 3:
        chi2 = float((((obs - exp) ** 2) / np.where(exp > 0, exp, 1))[keep].sum())
        assert chi2 < stats.chi2(int(keep.sum()) - 1).ppf(1 - 1e-6), (chi2, int(keep.sum()))


def _twin_ranges(ranges):
    out = []
    for r in ranges:
        out.append({"start": int(r.start), "stop": int(r.stop), "k": int(r.k), "types": [int(r.types[j]) for j in range(r.n_types)],
                    "thr": [int(r.cdf_thr[j]) for j in range(r.n_types)],
                    "min_len": {t: int(r.min_len[t]) for t in range(8)}, "max_len": {t: int(r.max_len[t]) for t in range(8)}})
    return out


def fast_plan_host(eng, L, ranges, key, seq):
    lib = _lib()
    arr = (_ffi.Range * len(ranges))(*ranges)
    n_recs, pool_len, empty = C.c_uint64(), C.c_uint64(), C.c_int()
    rc = lib.msim_dbg_fast_plan(eng.h, L, arr, len(ranges), key, seq, None, 0, None, 0, C.byref(n_recs), C.byref(pool_len), C.byref(empty))
    eng._check(rc)
    recs = np.zeros(n_recs.value, dtype=_ffi.RECORD_DTYPE)
    pool = np.zeros(max(pool_len.value, 1), dtype=np.uint8)
    rc = lib.msim_dbg_fast_plan(eng.h, L, arr, len(ranges), key, seq, C.c_void_p(recs.ctypes.data), len(recs), C.c_void_p(pool.ctypes.data),
                                len(pool), C.byref(n_recs), C.byref(pool_len), C.byref(empty))
    eng._check(rc)
    return recs, pool[:pool_len.value], bool(empty.value)


def assert_plan_equals_twin(recs, pool, empty, twin):
    t_recs, t_pool, t_empty = twin
    assert len(recs) == len(t_recs)
    if len(recs):
        P, S, X, T, A = (np.array(c, dtype=np.int64) for c in zip(*t_recs))
        assert np.array_equal(recs["pos"], P) and np.array_equal(recs["stop"], S) and np.array_equal(recs["type"], T)
        assert np.array_equal(recs["aux"], A) and np.array_equal(recs["extra"], X)
    assert bytes(pool) == t_pool
    assert empty == t_empty


def _blocks(eng):
    return {t: int(eng_params_block(eng)[t]) for t in range(1, 8)}


def eng_params_block(eng):
    return eng._params_block


def _mk(blocks=None, titv=2.0):
    eng = _engine(blocks, titv=titv)
    b = [1] * 8
    for t, v in (blocks or {}).items():
        b[t] = v
    eng._params_block = b
    import mutation_simulator_amd.mutator as mm
    p_ti = titv * (1 / (titv + 1))
    eng._ti_lim = min(mm._floor_scaled(p_ti) + 1, 1 << 53)
    return eng


SHAPES = {
    "snp_one_range": (400_000, None, lambda L: [_range(0, L - 1, 0.01, {1: 1.0}, {})]),
    "snp_tiny": (3_000, None, lambda L: [_range(10, L - 5, 0.004, {1: 1.0}, {})]),
    "snp_d3": (300_000, {t: 3 for t in range(1, 8)}, lambda L: [_range(0, L - 1, 0.02, {1: 1.0}, {})]),
    "snp_ranges_hot_cold": (600_000, None, lambda L: [_range(1_000, 99_999, 0.01, {1: 1.0}, {}), _range(120_000, 121_999, 0.4, {1: 1.0}, {}),
                                                      _range(122_000, 122_100, 0.45, {1: 1.0}, {}), _range(200_000, L - 1, 0.0007, {1: 1.0}, {})]),
    "svmix_one_range": (500_000, None, lambda L: [_range(0, L - 1, 0.008, C3_CHANCES, C3_LENS)]),
    "svmix_dense_end": (60_000, None, lambda L: [_range(0, L - 1, 0.05, {1: 0.01, 3: 0.01, 4: 0.01, 5: 0.02}, {3: (100, 900), 4: (200, 800), 5: (300, 1500)})]),
    "sn_block_7": (300_000, {1: 7}, lambda L: [_range(0, L - 1, 0.2, {1: 1.0}, {})]),
    "sn_block_svmix": (300_000, {1: 5, 2: 2, 3: 4}, lambda L: [_range(0, 149_999, 0.05, C3_CHANCES, C3_LENS), _range(150_000, L - 1, 0.1, {1: 0.9, 2: 0.1}, {2: (3, 9)})]),
    "rmt_own_settings_visit": (400_000, None, lambda L: [_range(0, 49_999, 0.02, {3: 0.01, 1: 0.01}, {3: (2000, 9000)}, order=[3, 1]),
                                                         _range(50_000, 50_400, 0.1, {1: 1.0}, {}),
                                                         _range(50_401, 52_000, 0.05, {1: 0.02, 4: 0.03}, {4: (500, 3000)}),
                                                         _range(52_001, 199_999, 0.01, C3_CHANCES, C3_LENS),
                                                         _range(200_000, L - 1, 0.004, {5: 0.002, 2: 0.002}, {5: (50, 4000), 2: (1, 200)}, order=[5, 2])]),
    "long_deletions_dependent_blocks": (2_000_000, None, lambda L: [_range(0, L - 1, 0.01, {1: 0.009, 3: 0.001}, {3: (50_000, 400_000)})]),
}


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_emulated_plan_equals_twin(name):
    L, blocks, mk = SHAPES[name]
    eng = _mk(blocks)
    ranges = mk(L)
    for key, seq in ((0xC0FFEE1234, 0), (7, 5)):
        recs, pool, empty = fast_plan_host(eng, L, ranges, key, seq)
        twin = ft.plan(L, _twin_ranges(ranges), {t: eng._params_block[t] for t in range(1, 8)}, eng._ti_lim, key, seq)
        assert_plan_equals_twin(recs, pool, empty, twin)
        if name.startswith("snp"):
            assert len(recs) == sum(int(r.k) for r in ranges)
    eng.close()


def test_fast_refusals_and_value_error():
    eng = _mk()
    L = 100_000
    with pytest.raises(ValueError, match="Sample larger than population"):
        fast_plan_host(eng, L, [_range(0, 999, 0.7, {1: 1.0}, {})], 1, 0)            # k > n: the reference's ValueError
    with pytest.raises(_ffi.MsimUnsupported, match="translocations"):
        fast_plan_host(eng, L, [_range(0, L - 1, 0.01, {1: 0.5, 6: 0.25, 7: 0.25}, {6: (5, 50)})], 1, 0)
    with pytest.raises(_ffi.MsimUnsupported, match="overlapping"):
        fast_plan_host(eng, L, [_range(0, 5000, 0.01, {1: 1.0}, {}), _range(4000, 9000, 0.01, {1: 1.0}, {})], 1, 0)
    eng.close()
