"""CPU tier: the statistical model behind the SNP sampler's anchored windows (plan_gpu.hip: enqueue_sample_ahead).

The host lays out the interval in which a sample planned ahead of the stream-position chain expects its start from the MEAN and
VARIANCE of the CPython-stream words every stage in front of it consumes (csrc/plan_gpu.h: sample_words_moments,
snp_words_moments).  A biased mean or an under-estimated variance would not give wrong results -- a start outside the interval
raises the overflow error and the contig is re-planned on the host -- but it would turn that recovery from a once-in-10^15 event
into a routine one.  So the formulas are held against what the reference's own draws do:

* ``random.sample(range(n), k)`` (``util.py:104``) through a ``random.Random`` that counts its ``getrandbits`` calls -- every
  call of the set path is one 32-bit word of the stream;
* the SNP draws of ``mutator.py:428-463``: ``random.uniform`` (2 words) and, for a transversion, ``random.randint(0, 1)``
  (``_randbelow(2)``: ``getrandbits(2)`` until < 2), counted the same way.
"""
from __future__ import annotations

import ctypes as C
import math
import random
import statistics

import pytest

from mutation_simulator_amd import _ffi


class _Counting(random.Random):
    """CPython's generator with a call counter on the one primitive the set path and randint draw words through."""

    def __init__(self, seed):
        super().__init__(seed)
        self.words = 0

    def getrandbits(self, k):
        self.words += (k + 31) // 32
        return super().getrandbits(k)

    def random(self):                       # uniform(): one 53-bit double = two words (genrand_res53)
        self.words += 2
        return super().random()


def _moments(n, k, K, ti_lim):
    lib = _ffi.load()
    fn = lib.msim_dbg_stream_moments
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
    out = (C.c_double * 4)()
    assert fn(n, k, K, ti_lim, out) == 0
    return tuple(out)


# (tolerances: the mean of R runs has standard error sqrt(Var / R); the sample variance of R runs of a near-normal sum has relative
#  standard error sqrt(2 / (R - 1)) -- 0.13 for R = 120; both asserted at about four of their standard errors)
@pytest.mark.parametrize("n,k", [
    (1_000_000, 10_000),          # 20-bit draws, accept ratio 0.95
    (1_048_577, 20_000),          # n = 2^20 + 1: 21-bit draws, accept ratio 0.5 -- the variance is mostly the rejections'
    (300_000, 60_000),            # k/n = 0.2: 6 700 duplicates, the variance mostly theirs
    (2_500_000, 25_000),          # the benchmark's rate
])
def test_sample_words_mean_and_variance_vs_random_sample(n, k):
    assert n - k > 0 and k > 5
    runs = 120
    words = []
    for s in range(runs):
        r = _Counting(1000 + s)
        got = r.sample(range(n), k)
        assert len(got) == k
        words.append(r.words)
    e, v, _, _ = _moments(n, k, 0, 0)
    mean, var = statistics.mean(words), statistics.variance(words)
    assert abs(mean - e) <= 4.0 * math.sqrt(v / runs) + 1.0, (mean, e, v)
    assert 0.55 * v <= var <= 1.6 * v, (var, v)


@pytest.mark.parametrize("titv", [0.0, 0.5, 2.0, 1e9])
def test_snp_words_mean_and_variance_vs_the_reference_draws(titv):
    from mutation_simulator_amd import mutator as mm

    class S:
        pass
    from mutation_simulator_amd.mut_types import MutType
    S.mut_block = {t: 1 for t in MutType}
    S.titv = titv
    ti_lim = int(mm.params_descriptor(S).ti_lim)
    p_ti = titv * (1 / (titv + 1))                      # mutator.py:436
    K, runs = 20_000, 120
    words = []
    for s in range(runs):
        r = _Counting(77 + s)
        for _ in range(K):                              # mutator.py:438-455: p = uniform(0, 1); p <= p_ti: transition; else randint(0, 1)
            if not (r.uniform(0, 1) <= p_ti):
                r.randint(0, 1)
        words.append(r.words)
    _, _, e, v = _moments(10, 1, K, ti_lim)
    mean = statistics.mean(words)
    assert abs(mean - e) <= 4.0 * math.sqrt(max(v, 1.0) / runs) + 1.0, (mean, e, v)
    if v >= 1.0:
        assert 0.55 * v <= statistics.variance(words) <= 1.6 * v
    else:
        assert statistics.variance(words) <= 4.0       # titv ~ inf: (almost) every SNP a transition, exactly 2 words each


def test_interval_of_a_typical_step_is_small_against_its_samples():
    """What the formulas mean for the 3 Gb benchmark: after 23 contigs (1 - 2.5 M SNPs each) the start of the 24th is known to
    +- 8 sigma = a few tens of thousands of words -- against samples of 0.6 - 4 M words."""
    v_total = 0.0
    for L in [248_000_000, 242_000_000, 198_000_000, 190_000_000, 181_000_000, 171_000_000, 159_000_000, 145_000_000] * 3:
        k = L // 100
        _, v, _, v2 = _moments(L - k, k, k, 6004799503160662)      # titv 2.0
        v_total += v + v2
    assert 8.0 * math.sqrt(v_total) < 120_000
