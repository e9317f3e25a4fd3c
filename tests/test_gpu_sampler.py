"""GPU sampler (PLAN on the device) vs the sequential host planner and the oracle.

The GPU sampler must land on the same records AND leave both MT19937 streams at the same position
as a sequential walk of the reference's draws -- including across contigs, where one off-by-one in
a stream cut would shift every later draw."""
from __future__ import annotations

import numpy as np
import pytest

from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm

pytestmark = pytest.mark.gpu


def _params(blocks=None, titv=1.0):
    class S:
        pass
    from mutation_simulator_amd.mut_types import MutType
    S.mut_block = {t: 1 for t in MutType}
    if blocks:
        for t, v in blocks.items():
            S.mut_block[MutType[t]] = v
    S.titv = titv
    return mm.params_descriptor(S)


def _snp_range(start, stop, k, token_order=False):
    r = _ffi.Range()
    r.start, r.stop, r.k = start, stop, k
    r.setsize = mm.sample_setsize(k)
    if token_order:                         # RMT line "in 0 ... sn x": chances IN 0.0, SN 1.0
        r.n_types = 2
        r.types[0], r.types[1] = 2, 1
        r.cdf_thr[0], r.cdf_thr[1] = 0, 1 << 53
        r.min_len[2], r.max_len[2] = 1, 2
    else:                                   # ARGS order SN, IN, DE, IV, DU, TL, TLI with p = 1, 0, ...
        r.n_types = 7
        for j, t in enumerate([1, 2, 3, 5, 4, 6, 7]):
            r.types[j] = t
            r.cdf_thr[j] = 1 << 53
        for t in (2, 3, 4, 6):
            r.min_len[t], r.max_len[t] = 1, 2
        r.min_len[5], r.max_len[5] = 2, 3
    return r


def _run(flags, contigs, params, seed=(42, 42)):
    """contigs: list of (length, [ranges]).  Returns per-contig records, final states, stats."""
    eng = _ffi.Engine(0, flags)
    eng.seed(*seed)
    eng.set_params(params)
    out = []
    for L, ranges in contigs:
        cid = eng.add_contig_synthetic(L, 7)
        eng.plan_contig(cid, ranges)
        recs, pool = eng.fetch_records(cid)
        out.append((recs.copy(), eng.plan_was_empty(cid), pool.copy()))
    st = eng.stats()
    states = [eng.get_mt_state(0), eng.get_mt_state(1)]
    eng.close()
    return out, states, st


def _compare(contigs, params, seed=(42, 42), host_chain=False):
    host, hs, hst = _run(_ffi.PLAN_HOST, contigs, params, seed)
    gpu, gs, gst = _run(_ffi.PLAN_GPU, contigs, params, seed)
    for (hr, he, hpool), (gr, ge, gpool) in zip(host, gpu):
        assert he == ge
        assert hr.shape == gr.shape
        assert np.array_equal(hr["pos"], gr["pos"])
        assert np.array_equal(hr["type"], gr["type"])
        assert np.array_equal(hr["stop"], gr["stop"])
        assert np.array_equal(hr["aux"], gr["aux"])
        assert np.array_equal(hr.view(np.uint8), gr.view(np.uint8))
        assert np.array_equal(hpool, gpool)
    assert hst["py_words"] == gst["py_words"] and hst["np_words"] == gst["np_words"]
    for (hm, hp), (gm, gp) in zip(hs, gs):
        # same stream position: the next outputs agree (the 624-word windows may be cut differently)
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)
    assert gst["plan_gpu_ms"] > 0 and (host_chain or gst["plan_host_ms"] == 0)
    return gst


def _next_words(mt, pos, n):
    import random
    r = random.Random()
    r.setstate((3, tuple(int(x) for x in mt) + (int(pos),), None))
    return [r.getrandbits(32) for _ in range(n)]


def test_c1_shape_matches_survey_known_answer():
    """1 Mb, k = 10 000: first positions 96, 178, 195 ..., 10 630 sample words."""
    out, states, st = _run(_ffi.PLAN_GPU, [(1_000_000, [_snp_range(0, 999_999, 10_000)])], _params())
    recs = out[0][0]
    assert list(recs["pos"][:6]) == [96, 178, 195, 208, 305, 430] and recs["pos"][-1] == 999_998
    assert len(recs) == 10_000 and np.all(recs["type"] == 1) and np.array_equal(recs["pos"], recs["stop"])


@pytest.mark.parametrize("L,k,titv", [
    (1_000_000, 10_000, 1.0), (5_000_000, 50_000, 2.0), (3_000_000, 150_000, 0.0),
    (1_400_000, 300_000, 1e9),            # k/n = 0.27: thousands of duplicate draws, multi-round tail
    (2_000_000, 340_000, 0.3),
    (4_194_304 + 41_000, 41_000, 0.5),    # n = 2^22 exactly -> 23-bit draws, accept ratio ~ 0.5
    (4_194_304 + 40_999, 41_000, 2.0),    # n = 2^22 - 1   -> 22-bit draws, accept ratio ~ 1.0
])
def test_single_range_vs_host(L, k, titv):
    assert (L - k) > mm.sample_setsize(k)       # CPython set path (pool path belongs to the host planner)
    _compare([(L, [_snp_range(0, L - 1, k)])], _params(titv=titv))


def test_contig_chain_and_multi_range_vs_host():
    contigs = [(3_000_000, [_snp_range(0, 2_999_999, 30_000)]),
               (1_000_000, []),                                            # nothing drawn
               (2_500_000, [_snp_range(0, 999_999, 10_000), _snp_range(1_000_000, 2_499_999, 120_000, True)]),
               (700_000, [_snp_range(100_000, 650_000, 5_000)])]
    _compare(contigs, _params(titv=2.0), seed=(7, 9))


def test_min_distance_three_vs_host():
    blocks = {t: 3 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    _compare([(2_000_000, [_snp_range(0, 1_999_999, 40_000)])], _params(blocks, titv=1.5), seed=(3, 4))


def test_ineligible_structures_are_refused_by_forced_gpu_mode():
    eng = _ffi.Engine(0, _ffi.PLAN_GPU)
    eng.seed(1, 1)
    cid = eng.add_contig_synthetic(1_000_000, 1)
    eng.set_params(_params())
    with pytest.raises(_ffi.MsimUnsupported):     # overlapping ranges (dict semantics) -> host planner territory
        eng.plan_contig(cid, [_sv_range(0, 599_999, 8_000, C3_CHANCES, C3_LENS), _sv_range(400_000, 999_999, 8_000, C3_CHANCES, C3_LENS)])
    with pytest.raises(_ffi.MsimUnsupported):     # SV mix on a tiny range -> host planner territory
        eng.plan_contig(cid, [_sv_range(0, 9_999, 100, C3_CHANCES, C3_LENS)])
    eng.close()


def test_auto_mode_mixes_engines_along_one_stream():
    """AUTO: GPU sampler for the big SNP ranges, host planner for the rest -- one continuous stream."""
    def sv_range(L):
        r = _snp_range(0, L - 1, int(L * 0.004))
        for j, thr in enumerate([0.5, 0.75, 1.0, 1.0, 1.0, 1.0, 1.0]):
            r.cdf_thr[j] = int(thr * (1 << 53))
        for t in (2, 3, 4):
            r.min_len[t], r.max_len[t] = 1, 20
        return r
    contigs = [(2_000_000, [_snp_range(0, 1_999_999, 20_000)]), (500_000, [sv_range(500_000)]),
               (1_500_000, [_snp_range(0, 1_499_999, 15_000)]), (300_000, [_snp_range(0, 299_999, 3_000)])]
    host, hs, hst = _run(_ffi.PLAN_HOST, contigs, _params(titv=2.0))
    auto, as_, ast = _run(_ffi.PLAN_AUTO, contigs, _params(titv=2.0))
    for (hr, _, hpool), (ar, _, apool) in zip(host, auto):
        assert np.array_equal(hr.view(np.uint8), ar.view(np.uint8))
        assert np.array_equal(hpool, apool)
    for (hm, hp), (am, ap) in zip(hs, as_):
        assert _next_words(hm, hp, 8) == _next_words(am, ap, 8)
    assert ast["plan_gpu_ms"] > 0 and ast["plan_host_ms"] > 0


# ---------------------------------------------------------------------- host-sampled contigs (RMT shape)
def _rmt_like_ranges(L, rng, n_blocks, hot=(), token_order=True):
    """Gene-blocking layout: `n_blocks` blocked stretches, std `sn 0.01` in the gaps, optional hot spots
    (start, stop, rate).  Returns drawing ranges in position order."""
    cuts = np.sort(rng.choice(np.arange(1, L - 1), size=2 * n_blocks, replace=False))
    ranges, at = [], 0
    for a, b in zip(cuts[0::2], cuts[1::2]):          # [at, a-1] drawing, [a, b] blocked
        if a - 1 > at:
            ranges.append((at, int(a) - 1, 0.01))
        at = int(b) + 1
    if at < L - 1:
        ranges.append((at, L - 1, 0.01))
    out = []
    for s, e, rate in sorted(list(ranges) + list(hot)):
        k = int(((e - s) + 1) * rate)
        if k:
            out.append(_snp_range(s, e, k, token_order))
    return out


@pytest.mark.parametrize("L,n_blocks,seed", [(3_000_000, 60, 1), (2_000_000, 400, 2), (5_000_000, 1500, 3)])
def test_many_small_ranges_vs_host(L, n_blocks, seed):
    rng = np.random.RandomState(seed)
    ranges = _rmt_like_ranges(L, rng, n_blocks)
    assert len(ranges) > n_blocks // 2
    _compare([(L, ranges)], _params(titv=2.0), seed=(seed, seed + 1), host_chain=True)


def test_pool_path_hot_spots_and_tiny_ranges_vs_host():
    """Hot spots whose sample takes CPython's pool path (n <= setsize), k <= 5 (setsize 21), k = 1, and a
    range that fills its whole population (k == n)."""
    ranges = [_snp_range(0, 99_999, 1_000), _snp_range(100_000, 100_999, 211),       # 1 kb at 0.211: pool path
              _snp_range(101_000, 101_024, 5), _snp_range(101_100, 101_120, 1), _snp_range(101_200, 101_209, 5),
              _snp_range(102_000, 901_999, 8_000, True), _snp_range(902_000, 902_099, 40)]
    for r in ranges[1:5]:
        d = 1
        assert (r.stop - (r.k - 1) * d) - r.start <= r.setsize                   # really the pool path
    _compare([(1_000_000, ranges)], _params(titv=0.7), seed=(8, 8), host_chain=True)
    blocks = {t: 2 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    _compare([(1_000_000, ranges[:2] + ranges[5:6])], _params(blocks, titv=1.0), seed=(9, 1), host_chain=True)   # d = 2


def test_host_sampled_chain_with_other_engines():
    rng = np.random.RandomState(5)
    sv = lambda L, rate: _sv_range(0, L - 1, int(L * rate), C3_CHANCES, C3_LENS)
    contigs = [(2_000_000, _rmt_like_ranges(2_000_000, rng, 100)), (2_000_000, [_snp_range(0, 1_999_999, 20_000)]),
               (1_500_000, [sv(1_500_000, 0.008)]), (1_000_000, _rmt_like_ranges(1_000_000, rng, 300, token_order=False)),
               (400_000, [sv(400_000, 0.005)]), (700_000, _rmt_like_ranges(700_000, rng, 20))]
    host, hs, hst = _run(_ffi.PLAN_HOST, contigs, _params(titv=2.0), (2, 3))
    auto, as_, ast = _run(_ffi.PLAN_AUTO, contigs, _params(titv=2.0), (2, 3))
    for (hr, he, hpool), (ar, ae, apool) in zip(host, auto):
        assert he == ae and np.array_equal(hr.view(np.uint8), ar.view(np.uint8)) and np.array_equal(hpool, apool)
    assert hst["py_words"] == ast["py_words"] and hst["np_words"] == ast["np_words"]
    for (hm, hp), (am, ap) in zip(hs, as_):
        assert _next_words(hm, hp, 8) == _next_words(am, ap, 8)


# ---------------------------------------------------------------------- SV mixes on the device
# (sample, type draw, SNP filter, records, insert pool and SNP draws on the GPU; only the boundary chain
#  over the non-SNP candidates on the host)
ARGS_ORDER = [1, 2, 3, 5, 4, 6, 7]          # SN, IN, DE, IV, DU, TL, TLI  (rmt.py:443-450)


def _sv_range(start, stop, k, chances, lens, order=ARGS_ORDER):
    """chances: {type id: p}; lens: {type id: (min, max)} -- thresholds built like mutator.range_descriptor."""
    r = _ffi.Range()
    r.start, r.stop, r.k = start, stop, k
    r.setsize = mm.sample_setsize(k)
    p = np.array([chances.get(t, 0.0) for t in order], dtype=np.float64)
    cdf = np.cumsum(p / p.sum())
    cdf /= cdf[-1]
    r.n_types = len(order)
    for j, t in enumerate(order):
        r.types[j] = t
        r.cdf_thr[j] = mm._ceil_scaled(float(cdf[j]))
    for t in (2, 3, 4, 6):
        r.min_len[t], r.max_len[t] = lens.get(t, (1, 2))
    r.min_len[5], r.max_len[5] = lens.get(5, (2, 3))
    return r


C3_CHANCES = {1: 0.005, 2: 0.001, 3: 0.001, 4: 0.0005, 5: 0.0005}
C3_LENS = {2: (1, 50), 3: (1, 50), 4: (50, 500), 5: (50, 500)}


@pytest.mark.parametrize("L,rate,chances,lens,blocks,titv", [
    (3_000_000, 0.008, C3_CHANCES, C3_LENS, None, 1.0),                   # BASELINE configs[2] shape
    (1_000_000, 0.05, C3_CHANCES, C3_LENS, None, 2.0),                    # dense: most candidates blocked
    (2_000_000, 0.01, {2: 1.0}, {2: (1, 1)}, None, 1.0),                  # insertions only, width-1 randint
    (2_000_000, 0.01, {3: 0.5, 1: 0.5}, {3: (1, 3000)}, None, 0.0),       # long deletions swallow many candidates
    (2_500_000, 0.006, {1: 0.2, 4: 0.4, 5: 0.4}, {4: (2, 2000), 5: (2, 2000)}, {"DU": 40, "IV": 7}, 1e9),
    (2_000_000, 0.02, C3_CHANCES, C3_LENS, {t: 3 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}, 0.5),
    (600_000, 0.01, {1: 0.1, 5: 0.9}, {5: (2, 400_000)}, None, 1.0),      # IV reach beyond the contig end: dropped undrawn
    (600_000, 0.01, {1: 0.5, 3: 0.25, 4: 0.25}, {3: (1, 500_000), 4: (1, 500_000)}, None, 1.0),   # DE/DU clamped at the end
])
def test_sv_mix_vs_host(L, rate, chances, lens, blocks, titv):
    k = int(L * rate)
    _compare([(L, [_sv_range(0, L - 1, k, chances, lens)])], _params(blocks, titv=titv), host_chain=True)


@pytest.mark.parametrize("case", range(16))
def test_sv_mix_random_sweep_vs_host(case):
    """Seeded random SV-mix settings (type subset and chances, length ranges, blocks, minimum distance, range placement,
    titv): the device engine must reproduce the host planner's records, insert pool and stream positions."""
    rs = np.random.RandomState(1000 + case)
    L = int(rs.randint(700_000, 2_500_000))
    start = int(rs.choice([0, rs.randint(1, L // 4)]))
    stop = int(rs.choice([L - 1, L - 1 - rs.randint(1, L // 4)]))
    types = [1] + [t for t in (2, 3, 4, 5) if rs.rand() < 0.7]
    if len(types) == 1:
        types.append(int(rs.choice([2, 3, 4, 5])))
    chances = {t: float(rs.uniform(0.05, 1.0)) for t in types}
    lens = {}
    for t in (2, 3, 4):
        a = int(rs.randint(1, 40))
        lens[t] = (a, a + int(rs.choice([0, 1, 7, 60, 700, 5000])))
    a = int(rs.randint(2, 40))
    lens[5] = (a, a + int(rs.choice([0, 1, 7, 60, 700, 5000])))
    dmin = int(rs.choice([1, 1, 2, 4]))
    blocks = {name: dmin + int(rs.choice([0, 0, 1, 5, 50])) for name in ("IN", "DE", "IV", "DU", "TL", "TLI")}
    blocks["SN"] = dmin                                      # SNP block == sampling distance (device engines' precondition)
    blocks[str(rs.choice(["IN", "DE", "IV", "DU"]))] = dmin  # make sure the minimum is dmin
    rate = float(rs.choice([0.006, 0.01, 0.02, 0.04]))
    k = int((stop - start + 1) * rate)
    order = list(ARGS_ORDER) if rs.rand() < 0.5 else [int(x) for x in rs.permutation(types)]
    r = _sv_range(start, stop, k, chances, lens, order=order)
    _compare([(L, [r])], _params(blocks, titv=float(rs.choice([0.0, 0.5, 1.0, 2.0, 1e9]))), seed=(case, 2 * case + 1),
             host_chain=True)


@pytest.mark.parametrize("case", range(8))
def test_host_sampled_random_sweep_vs_host(case):
    """Seeded random range layouts (counts, sizes from a handful of bases to megabases, rates up to pool-path density,
    minimum distance) for the host-sampled engine."""
    rs = np.random.RandomState(2000 + case)
    L = int(rs.randint(500_000, 4_000_000))
    d = int(rs.choice([1, 1, 2, 5]))
    ranges, at = [], int(rs.randint(0, 1000))
    while at < L - 50:
        length = int(min(L - at, rs.choice([8, 30, 200, 2000, 20_000, 300_000, 1_500_000])))
        rate = float(rs.choice([0.0005, 0.01, 0.03, 0.1, 0.2]))
        k = int(length * rate)
        n = (at + length - 1 - (k - 1) * d) - at
        if k > 0 and n >= k:
            ranges.append(_snp_range(at, at + length - 1, k, bool(rs.rand() < 0.5)))
        at += length + int(rs.choice([1, 1, 50, 5000]))
    assert ranges
    blocks = {t: d for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    _compare([(L, ranges)], _params(blocks, titv=float(rs.choice([0.0, 1.0, 2.0]))), seed=(case + 7, case + 9),
             host_chain=True)


def test_sv_mix_rmt_token_order_and_inner_range():
    """RMT-style chance order (DU, SN, IN) on a range that does not start at 0 nor end at the contig end."""
    r = _sv_range(200_000, 1_799_999, 16_000, {4: 0.1, 1: 0.6, 2: 0.3}, {4: (5, 90), 2: (1, 12)}, order=[4, 1, 2])
    _compare([(2_000_000, [r])], _params(titv=2.0), seed=(11, 5), host_chain=True)


def test_sv_mix_contig_chain():
    """SNP-only sampler, SV-mix path and host planner alternate along one pair of streams."""
    sv = lambda L, rate: _sv_range(0, L - 1, int(L * rate), C3_CHANCES, C3_LENS)
    contigs = [(2_000_000, [_snp_range(0, 1_999_999, 20_000)]), (3_000_000, [sv(3_000_000, 0.008)]),
               (1_000_000, [sv(1_000_000, 0.02)]), (300_000, [sv(300_000, 0.008)]),      # k < 4096: host planner
               (1_500_000, [_snp_range(0, 1_499_999, 15_000)]), (2_000_000, [sv(2_000_000, 0.004)])]
    host, hs, hst = _run(_ffi.PLAN_HOST, contigs, _params(titv=2.0), (5, 6))
    auto, as_, ast = _run(_ffi.PLAN_AUTO, contigs, _params(titv=2.0), (5, 6))
    for (hr, he, hpool), (ar, ae, apool) in zip(host, auto):
        assert he == ae and np.array_equal(hr.view(np.uint8), ar.view(np.uint8)) and np.array_equal(hpool, apool)
    assert hst["py_words"] == ast["py_words"] and hst["np_words"] == ast["np_words"]
    for (hm, hp), (am, ap) in zip(hs, as_):
        assert _next_words(hm, hp, 8) == _next_words(am, ap, 8)


# ---------------------------------------------------------------------- host-chain engine (several settings per contig)
# (samples AND boundary passes walked by the host in one go over device-made words, accept tables and candidate types;
#  SNP filter with per-range reset, visit filter across range borders, records, pool and SNP draws on the device)
def _gene_gaps(L, rs, n_blocks):
    cuts = np.sort(rs.choice(np.arange(1, L - 1), size=2 * n_blocks, replace=False))
    out, at = [], 0
    for a, b in zip(cuts[0::2], cuts[1::2]):
        if a - 1 > at:
            out.append((at, int(a) - 1))
        at = int(b) + 1
    if at < L - 1:
        out.append((at, L - 1))
    return out


def _rate_range(s, e, rate, chances, lens, order=None):
    order = order or [t for t in ARGS_ORDER if t in chances]
    return _sv_range(s, e, int(((e - s) + 1) * rate), chances, lens, order=order)


@pytest.mark.parametrize("L,n_blocks,seed", [(3_000_000, 150, 1), (2_000_000, 600, 2), (6_000_000, 80, 3)])
def test_gene_blocks_with_sv_std_vs_host(L, n_blocks, seed):
    """The mainstream RMT shape: blocked genes, one SV `std` settings object in every gap."""
    rs = np.random.RandomState(seed)
    ranges = [_rate_range(s, e, 0.008, C3_CHANCES, C3_LENS) for s, e in _gene_gaps(L, rs, n_blocks)]
    ranges = [r for r in ranges if r.k]
    st = _compare([(L, ranges)], _params(titv=2.0), seed=(seed, seed + 5), host_chain=True)
    assert st["contigs_hostchain"] == 1 and st["contigs_host"] == 0


def test_host_chain_spans_cross_range_borders_vs_host():
    """Touching ranges with long DE / DU / IV: spans of range i swallow candidates of range i+1 (and of i+2 across a tiny
    range) -- they take part in their own boundary pass but are never visited (mutator.py:184, 376,386,398)."""
    L = 1_500_000
    ranges, at = [], 0
    rs = np.random.RandomState(4)
    while at < L - 10:
        length = int(rs.choice([60, 300, 2_000, 9_000, 40_000]))
        e = min(L - 1, at + length - 1)
        ranges.append(_rate_range(at, e, 0.03, {1: 0.4, 3: 0.25, 4: 0.2, 5: 0.1, 2: 0.05},
                                  {3: (200, 900), 4: (100, 700), 5: (50, 800), 2: (1, 9)}))
        at = e + 1
    ranges = [r for r in ranges if r.k]
    _compare([(L, ranges)], _params(titv=1.0), seed=(9, 9), host_chain=True)


def test_host_chain_hot_cold_own_lengths_pool_path_vs_host():
    L = 2_400_000
    blocks = {t: 2 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    blocks["DE"] = 5
    ranges = [
        _rate_range(0, 399_999, 0.01, {1: 1.0}, {}),
        _rate_range(400_000, 400_999, 0.2, {4: 0.1, 1: 0.8, 2: 0.1}, {4: (5, 8), 2: (1, 4)}, order=[4, 1, 2]),   # pool path
        _rate_range(401_000, 999_999, 0.008, C3_CHANCES, C3_LENS),
        _rate_range(1_000_000, 1_000_040, 0.1, {1: 0.5, 3: 0.5}, {3: (1, 50)}),                                # k = 4
        _rate_range(1_000_100, 1_799_999, 0.02, {1: 0.3, 3: 0.3, 5: 0.4}, {3: (1, 50), 5: (50, 500)}, order=[5, 3, 1]),
        _rate_range(1_800_000, 2_399_999, 0.004, {2: 1.0}, {2: (3, 3)}),                                      # width-1 randint
    ]
    assert (ranges[1].stop - (ranges[1].k - 1) * 2) - ranges[1].start <= ranges[1].setsize
    st = _compare([(L, ranges)], _params(blocks, titv=0.5), seed=(21, 22), host_chain=True)
    assert st["contigs_hostchain"] == 1


@pytest.mark.parametrize("sn_block", [2, 7])
def test_snp_block_above_sampling_distance_vs_host(sn_block):
    """sn_block > min(block): SNPs block their successors, every candidate chains -- SNP-only contigs and single large
    ranges included (the other device engines decline them)."""
    L = 2_000_000
    rs = np.random.RandomState(sn_block)
    ranges = []
    for s, e in _gene_gaps(L, rs, 60):
        kind = rs.randint(0, 3)
        if kind == 0:
            ranges.append(_rate_range(s, e, 0.05, {1: 1.0}, {}))
        elif kind == 1:
            ranges.append(_rate_range(s, e, 0.3, {1: 0.9, 2: 0.1}, {2: (1, 4)}))
        else:
            ranges.append(_rate_range(s, e, 0.02, {1: 0.2, 3: 0.4, 5: 0.2, 4: 0.2}, {3: (3, 60), 5: (5, 90), 4: (4, 80)}))
    ranges = [r for r in ranges if r.k and (r.stop - (r.k - 1)) - r.start >= r.k]
    contigs = [(L, ranges), (1_000_000, [_snp_range(0, 999_999, 10_000)]),
               (1_500_000, [_rate_range(0, 1_499_999, 0.008, C3_CHANCES, C3_LENS)])]
    st = _compare(contigs, _params({"SN": sn_block}, titv=1.7), seed=(3, 4), host_chain=True)
    assert st["contigs_hostchain"] == 3


@pytest.mark.parametrize("case", range(12))
def test_host_chain_random_layouts_vs_host(case):
    rs = np.random.RandomState(500 + case)
    L = int(rs.randint(400_000, 4_000_000))
    d = int(rs.choice([1, 1, 2, 4]))
    blocks = {name: d + int(rs.choice([0, 0, 1, 5, 40])) for name in ("IN", "DE", "IV", "DU", "TL", "TLI")}
    blocks["SN"] = d if rs.rand() < 0.7 else d + int(rs.randint(1, 4))
    blocks[str(rs.choice(["IN", "DE", "IV", "DU"]))] = d
    n_sets = int(rs.randint(1, 5))
    widths = [(1, 1 + int(rs.choice([0, 3, 49, 450]))) for _ in range(2)]          # at most 4 randint classes in total
    sets = []
    for _ in range(n_sets):
        types = [1] + [t for t in (2, 3, 4, 5) if rs.rand() < 0.6]
        chances = {t: float(rs.uniform(0.05, 1.0)) for t in types}
        lens = {}
        for t in (2, 3, 4):
            a, b = widths[int(rs.randint(0, 2))]
            off = int(rs.randint(0, 30))
            lens[t] = (a + off, b + off)
        a, b = widths[int(rs.randint(0, 2))]
        lens[5] = (a + 1, b + 1)
        sets.append((chances, lens, [int(x) for x in rs.permutation(types)], float(rs.choice([0.002, 0.01, 0.03, 0.1]))))
    ranges, at = [], int(rs.randint(0, 500))
    while at < L - 50:
        length = int(min(L - at, rs.choice([8, 30, 200, 2_000, 20_000, 150_000, 900_000])))
        chances, lens, order, rate = sets[int(rs.randint(0, n_sets))]
        r = _rate_range(at, at + length - 1, rate, chances, lens, order)
        if r.k > 0 and (r.stop - (r.k - 1) * d) - r.start >= r.k:
            ranges.append(r)
        at += length + int(rs.choice([0, 0, 1, 50, 3_000]))
    assert sum(r.k for r in ranges) >= 1024
    _compare([(L, ranges)], _params(blocks, titv=float(rs.choice([0.0, 1.0, 2.0]))), seed=(case + 1, case + 11), host_chain=True)


def test_host_chain_with_other_engines_along_one_stream():
    rs = np.random.RandomState(6)
    sv = lambda L, rate: _sv_range(0, L - 1, int(L * rate), C3_CHANCES, C3_LENS)
    gaps = lambda L, n: [r for r in (_rate_range(s, e, 0.008, C3_CHANCES, C3_LENS) for s, e in _gene_gaps(L, rs, n)) if r.k]
    contigs = [(2_000_000, gaps(2_000_000, 100)), (2_000_000, [_snp_range(0, 1_999_999, 20_000)]),
               (1_500_000, [sv(1_500_000, 0.008)]), (1_000_000, _rmt_like_ranges(1_000_000, rs, 300, token_order=False)),
               (900_000, gaps(900_000, 300)), (300_000, gaps(300_000, 10)), (700_000, _rmt_like_ranges(700_000, rs, 20))]
    host, hs, hst = _run(_ffi.PLAN_HOST, contigs, _params(titv=2.0), (2, 3))
    auto, as_, ast = _run(_ffi.PLAN_AUTO, contigs, _params(titv=2.0), (2, 3))
    for (hr, he, hpool), (ar, ae, apool) in zip(host, auto):
        assert he == ae and np.array_equal(hr.view(np.uint8), ar.view(np.uint8)) and np.array_equal(hpool, apool)
    assert hst["py_words"] == ast["py_words"] and hst["np_words"] == ast["np_words"]
    for (hm, hp), (am, ap) in zip(hs, as_):
        assert _next_words(hm, hp, 8) == _next_words(am, ap, 8)
    assert ast["contigs_hostchain"] >= 2 and ast["contigs_snp"] == 1 and ast["contigs_svmix"] == 1 and ast["contigs_hostcut"] == 2


TL_CHANCES = {1: 0.01, 2: 0.01, 3: 0.01, 5: 0.01, 4: 0.01, 6: 0.005, 7: 0.005}     # README: -sn -in -de -du -iv -tl 0.01 each
TL_LENS = {2: (10, 100), 3: (1, 2), 4: (1, 2), 6: (1, 2)}


@pytest.mark.parametrize("L,seed", [(3_000_000, 1), (1_200_000, 2), (80_000_000, 3)])
def test_translocations_readme_flags_vs_host(L, seed):
    """The reference's own benchmark flags (README "Performance": every type at 0.01, translocations included), one range
    per contig as ARGS mode makes them: the SV-mix engine -- device sample, TL / TLI on the host's boundary walk,
    __link_tls on a word window fetched behind it, TLI records with their linked spans from the device."""
    r = _sv_range(0, L - 1, int(L * 0.06), TL_CHANCES, TL_LENS)
    st = _compare([(L, [r])], _params(titv=1.0), seed=(seed, seed + 3), host_chain=True)
    assert st["contigs_svmix"] == 1


@pytest.mark.parametrize("chances,lens,order", [
    ({1: 0.5, 6: 0.1, 7: 0.4}, {6: (5, 60)}, [7, 1, 6]),                 # many more sites than spans: __fix_tl_amount deletes sites
    ({1: 0.3, 6: 0.45, 7: 0.05, 3: 0.2}, {6: (200, 900), 3: (200, 900)}, [6, 3, 7, 1]),   # ... deletes spans (which still blocked)
    ({1: 0.5, 7: 0.5}, {}, [1, 7]),                                      # sites only: nothing is drawn, start = pos, stop = 0
    ({1: 0.5, 6: 0.5}, {6: (1, 1)}, [1, 6]),                             # spans only, length 1: all deleted
    ({6: 0.5, 7: 0.5}, {6: (1, 3)}, [6, 7]),                             # no SNPs at all; lengths below 2 never invert
])
def test_translocations_sv_mix_shapes_vs_host(chances, lens, order):
    L = 2_000_000
    blocks = {t: 2 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    blocks["TL"] = 7
    contigs = [(L, [_rate_range(0, L - 1, 0.01, chances, lens, order=order)]),
               (700_000, [_rate_range(1_000, 650_000, 0.02, chances, lens, order=order)])]
    st = _compare(contigs, _params(blocks, titv=2.0), seed=(21, 22), host_chain=True)
    assert st["contigs_svmix"] == 2


def test_translocations_readme_flags_in_two_ranges_take_the_host_chain_engine():
    L = 2_400_000
    ranges = [_sv_range(0, L // 2 - 1, int(L * 0.03), TL_CHANCES, TL_LENS), _sv_range(L // 2, L - 1, int(L * 0.03), TL_CHANCES, TL_LENS)]
    st = _compare([(L, ranges)], _params(titv=1.0), seed=(5, 8), host_chain=True)
    assert st["contigs_hostchain"] == 1


def test_translocations_across_ranges_vs_host():
    L = 1_800_000
    blocks = {t: 2 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    blocks["TL"] = 9
    ranges = [
        _rate_range(0, 399_999, 0.02, {1: 0.5, 6: 0.1, 7: 0.4}, {6: (5, 60)}, order=[7, 1, 6]),
        _rate_range(400_000, 799_999, 0.01, {1: 1.0}, {}, order=[1]),
        _rate_range(800_000, 1_199_999, 0.02, {1: 0.3, 6: 0.45, 7: 0.05, 3: 0.2}, {6: (200, 900), 3: (200, 900)}, order=[6, 3, 7, 1]),
        _rate_range(1_200_000, 1_200_300, 0.1, {6: 0.5, 7: 0.5}, {6: (5, 60)}, order=[6, 7]),
        _rate_range(1_200_301, L - 1, 0.015, {1: 0.4, 2: 0.2, 6: 0.2, 7: 0.2}, {2: (1, 9), 6: (5, 60)}),
    ]
    contigs = [(L, ranges), (600_000, [_rate_range(0, 599_999, 0.01, {1: 0.5, 7: 0.5}, {}, order=[1, 7])]),       # sites without spans
               (900_000, [_sv_range(0, 899_999, 9_000, C3_CHANCES, C3_LENS)])]
    _compare(contigs, _params(blocks, titv=2.0), seed=(13, 14), host_chain=True)


def test_plan_chain_leaves_the_streams_where_plan_contig_does():
    """msim_plan_chain (multi-GPU: a rank that does not own a contig) through every engine -- SNP sampler, SV mix, host-cut,
    host-chain, host planner -- must advance both streams exactly like msim_plan_contig, alternating with owned contigs."""
    rs = np.random.RandomState(8)
    sv = lambda L, rate: _sv_range(0, L - 1, int(L * rate), C3_CHANCES, C3_LENS)
    gaps = lambda L, n: [r for r in (_rate_range(s, e, 0.008, C3_CHANCES, C3_LENS) for s, e in _gene_gaps(L, rs, n)) if r.k]
    contigs = [(2_000_000, [_snp_range(0, 1_999_999, 20_000)]), (1_500_000, [sv(1_500_000, 0.008)]),
               (1_000_000, _rmt_like_ranges(1_000_000, rs, 300)), (2_000_000, gaps(2_000_000, 100)),
               (300_000, [sv(300_000, 0.004)]), (900_000, [_snp_range(0, 899_999, 9_000)]), (1_200_000, gaps(1_200_000, 40))]
    full, fs, fst = _run(_ffi.PLAN_AUTO, contigs, _params(titv=2.0), (4, 5))
    for owner_mask in (0b0000000, 0b1010101, 0b0101010):
        eng = _ffi.Engine(0)
        eng.seed(4, 5)
        eng.set_params(_params(titv=2.0))
        for i, (L, ranges) in enumerate(contigs):
            if owner_mask >> i & 1:
                cid = eng.add_contig_synthetic(L, 7)
                eng.plan_contig(cid, ranges)
                recs, pool = eng.fetch_records(cid)
                assert np.array_equal(recs.view(np.uint8), full[i][0].view(np.uint8)) and np.array_equal(pool, full[i][2])
            else:
                eng.plan_chain(L, ranges)
        st = eng.stats()
        assert st["py_words"] == fst["py_words"] and st["np_words"] == fst["np_words"]
        for (hm, hp), stream in zip(fs, (0, 1)):
            gm, gp = eng.get_mt_state(stream)
            assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)
        assert sum(v for k, v in st.items() if k.startswith("contigs_")) == bin(owner_mask).count("1")
        eng.close()


def test_full_size_sv_mix_gpu_vs_host_planner():
    """BASELINE config 3 at full size (3 Gb, 24 contigs, 24 M candidates): records, insert pools and both
    final stream positions equal the sequential host planner's."""
    import hashlib

    import bench
    from test_gpu_fullsize import C3
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.workload_settings(lengths, snp=0.005, titv=1.0, extra=C3)
    params = mm.params_descriptor(sim)

    def run(flags):
        eng = _ffi.Engine(0, flags)
        eng.seed(42, 42)
        eng.set_params(params)
        sums = []
        for chrom in sim.chromosomes:
            cid = eng.add_contig_synthetic(lengths[chrom.number], 1)
            eng.plan_contig(cid, mm.plan_descriptors(chrom))
            recs, pool = eng.fetch_records(cid)
            sums.append((len(recs), len(pool), hashlib.sha256(recs.tobytes()).hexdigest(),
                         hashlib.sha256(pool.tobytes()).hexdigest()))
            eng.clear()
        states = [eng.get_mt_state(0), eng.get_mt_state(1)]
        st = eng.stats()
        eng.close()
        return sums, states, st

    hsum, hs, hst = run(_ffi.PLAN_HOST)
    gsum, gs, gst = run(_ffi.PLAN_GPU)
    assert hsum == gsum
    assert hst["py_words"] == gst["py_words"] and hst["np_words"] == gst["np_words"]
    for (hm, hp), (gm, gp) in zip(hs, gs):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)


def test_full_size_rmt_host_sampled_vs_host_planner():
    """BASELINE config 4 at full size (3 Gb, 24 contigs, ~41 k drawing ranges incl. pool-path hot spots,
    17.8 M SNPs): records and both final stream positions equal the sequential host planner's."""
    import hashlib

    import bench
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.workload_settings_rmt(lengths, bench.c4_rmt_text(lengths))
    params = mm.params_descriptor(sim)
    descs = [mm.plan_descriptors(ch) for ch in sim.chromosomes]
    assert sum(len(d) for d in descs) > 30_000

    def run(flags):
        eng = _ffi.Engine(0, flags)
        eng.seed(42, 42)
        eng.set_params(params)
        sums = []
        for chrom in sim.chromosomes:
            cid = eng.add_contig_synthetic(lengths[chrom.number], 1)
            eng.plan_contig(cid, descs[chrom.number])
            recs, pool = eng.fetch_records(cid)
            sums.append((len(recs), len(pool), hashlib.sha256(recs.tobytes()).hexdigest()))
            eng.clear()
        states = [eng.get_mt_state(0), eng.get_mt_state(1)]
        st = eng.stats()
        eng.close()
        return sums, states, st

    hsum, hs, hst = run(_ffi.PLAN_HOST)
    gsum, gs, gst = run(_ffi.PLAN_GPU)
    assert hsum == gsum
    assert hst["py_words"] == gst["py_words"] and hst["np_words"] == gst["np_words"]
    for (hm, hp), (gm, gp) in zip(hs, gs):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)


def test_full_size_genome_gpu_sampler_vs_host_planner():
    """BASELINE config 2 at full size (3 Gb, 24 contigs, 30 M SNPs): every record and both final
    stream positions equal the sequential host walk (> 200 M MT19937 words, > 1000 stream chunks,
    i.e. every jump-ahead level in use)."""
    import bench
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.workload_settings(lengths)
    params = mm.params_descriptor(sim)

    def run(flags):
        eng = _ffi.Engine(0, flags)
        eng.seed(42, 42)
        eng.set_params(params)
        sums = []
        for chrom in sim.chromosomes:
            cid = eng.add_contig_synthetic(lengths[chrom.number], 1)
            eng.plan_contig(cid, mm.plan_descriptors(chrom))
            recs, _ = eng.fetch_records(cid)
            sums.append((len(recs), int(recs["pos"].astype(np.uint64).sum()), int(recs["aux"].astype(np.uint64).sum()),
                         __import__("hashlib").sha256(recs.tobytes()).hexdigest()))
            eng.clear()
        states = [eng.get_mt_state(0), eng.get_mt_state(1)]
        st = eng.stats()
        eng.close()
        return sums, states, st

    hsum, hs, hst = run(_ffi.PLAN_HOST)
    gsum, gs, gst = run(_ffi.PLAN_GPU)
    assert hsum == gsum
    assert hst["py_words"] == gst["py_words"] and hst["np_words"] == gst["np_words"]
    for (hm, hp), (gm, gp) in zip(hs, gs):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)


def test_five_and_eight_randint_classes_vs_host():
    """Five SV types with five different length widths on one range (SV-mix engine) and eight widths over the settings of
    one contig (host-chain engine): 8 table slots per word position, 9-bit slot increments."""
    L = 3_000_000
    five = {3: (1, 9), 2: (1, 30), 5: (2, 120), 4: (5, 700), 6: (3, 50)}
    chances = {1: 0.3, 3: 0.15, 2: 0.15, 5: 0.1, 4: 0.1, 6: 0.12, 7: 0.08}
    st = _compare([(L, [_rate_range(0, L - 1, 0.01, chances, five)])], _params(titv=2.0), seed=(3, 4), host_chain=True)
    assert st["contigs_svmix"] == 1
    other = {3: (2, 70), 2: (1, 5), 5: (2, 14), 4: (5, 700), 6: (3, 50)}
    ranges = [_rate_range(0, 999_999, 0.01, chances, five), _rate_range(1_000_000, 1_999_999, 0.012, chances, other),
              _rate_range(2_000_000, L - 1, 0.01, {1: 0.5, 2: 0.5}, {2: (1, 30)})]
    st = _compare([(L, ranges)], _params({"DE": 3, "TL": 4}, titv=1.0), seed=(5, 6), host_chain=True)
    assert st["contigs_hostchain"] == 1


@pytest.mark.parametrize("n_contigs", [1, 3, 7, 8])
def test_plan_and_apply_a_whole_genome_then_ask_equals_contig_by_contig(n_contigs):
    """The fast order of the C-ABI -- plan + apply every contig, only then read -- lets the SNP sampler gather emission and
    APPLY in groups (gpu_emit_flush: one launch per stage for a pair of contigs, the last group partial, flushed by the first
    reader).  Records, insert-free mutated streams and both stream positions must equal the contig-by-contig order through
    the sequential host planner."""
    import hashlib
    rs = np.random.RandomState(100 + n_contigs)
    lengths = [int(x) for x in rs.randint(3_000_000, 9_000_000, size=n_contigs)]
    params = _params(titv=2.0)

    def ranges(L):
        return [_snp_range(0, L - 1, int(L * 0.01))]

    # contig by contig, host planner
    want = []
    eng = _ffi.Engine(0, _ffi.PLAN_HOST)
    eng.seed(7, 9)
    eng.set_params(params)
    for i, L in enumerate(lengths):
        cid = eng.add_contig_synthetic(L, 50 + i)
        eng.plan_contig(cid, ranges(L))
        eng.apply_contig(cid)
        recs, _ = eng.fetch_records(cid)
        want.append((hashlib.sha256(recs.tobytes()).hexdigest(), eng.result_checksum(cid)))
        eng.clear()
    want_states = [eng.get_mt_state(0), eng.get_mt_state(1)]
    eng.close()
    # everything enqueued first, on the device engines
    eng = _ffi.Engine(0, _ffi.PLAN_GPU)
    eng.seed(7, 9)
    eng.set_params(params)
    cids = [eng.add_contig_synthetic(L, 50 + i) for i, L in enumerate(lengths)]
    for cid, L in zip(cids, lengths):
        eng.plan_contig(cid, ranges(L))
        eng.apply_contig(cid)
    got = []
    for cid in cids:
        recs, _ = eng.fetch_records(cid)
        got.append((hashlib.sha256(recs.tobytes()).hexdigest(), eng.result_checksum(cid)))
    assert got == want
    assert eng.stats()["contigs_snp"] == n_contigs
    for (hm, hp), (gm, gp) in zip(want_states, [eng.get_mt_state(0), eng.get_mt_state(1)]):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)
    # ... and once more with one contig planned again in between (its group goes out first)
    eng.seed(7, 9)
    for j, (cid, L) in enumerate(zip(cids, lengths)):
        eng.plan_contig(cid, ranges(L))
        eng.apply_contig(cid)
        if j == 1:
            assert hashlib.sha256(eng.fetch_records(cid)[0].tobytes()).hexdigest() == want[1][0]
    assert [eng.result_checksum(cid) for cid in cids] == [w[1] for w in want]
    eng.close()
