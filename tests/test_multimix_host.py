"""The general host-chain PLAN engine (plan_contig_gpu_multimix: contigs whose drawing ranges have different
settings, SV types on many small ranges, SNP block above the sampling distance) on the CPU tier.

`msim_dbg_multimix_plan` runs the engine's algorithm end to end on the host -- the real host walk
(`multimix_walk_host`: sample -> positions -> boundary chain over accept tables, range after range along one
word window) plus a sequential restatement of what the device does around it (types by ordinal, accept tables,
clipped running maximum for the SNP filter, the visit filter across range borders, records).  It must arrive at
the host planner's record table, insert pool and stream positions; the host planner itself is held against the
reference's goldens and the oracle elsewhere (test_cabi_host.py, test_property_host.py)."""
from __future__ import annotations

import ctypes as C
import random

import numpy as np
import pytest

from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm

ARGS_ORDER = [1, 2, 3, 5, 4, 6, 7]


def _engine(blocks, titv=1.0, seed=(1, 2)):
    eng = _ffi.Engine(device=-1)
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    for t, v in (blocks or {}).items():
        p.block[t] = v
    p_ti = titv * (1 / (titv + 1))
    p.ti_lim = min(mm._floor_scaled(p_ti) + 1, 1 << 53)
    eng.set_params(p)
    eng.seed(*seed)
    return eng


def _range(start, stop, rate, chances, lens, order=None):
    order = order or [t for t in ARGS_ORDER if t in chances] or [1]
    r = _ffi.Range()
    r.start, r.stop = start, stop
    r.k = int(((stop - start) + 1) * rate)
    r.setsize = mm.sample_setsize(r.k)
    p = np.array([chances.get(t, 0.0) for t in order], dtype=np.float64)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    r.n_types = len(order)
    for j, t in enumerate(order):
        r.types[j] = t
        r.cdf_thr[j] = mm._ceil_scaled(float(cdf[j]))
    for t in (2, 3, 4, 6):
        r.min_len[t], r.max_len[t] = lens.get(t, (1, 2))
    r.min_len[5], r.max_len[5] = lens.get(5, (2, 3))
    return r


def _dbg(lib):
    lib.msim_dbg_multimix_plan.restype = C.c_int
    lib.msim_dbg_multimix_plan.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_ffi.Range), C.c_int, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_int)]
    return lib


def _next_words(mt, pos, n):
    r = random.Random()
    r.setstate((3, tuple(int(x) for x in mt) + (int(pos),), None))
    return [r.getrandbits(32) for _ in range(n)]


def _both(L, ranges, blocks, titv, seed, expect_unsupported=False):
    arr = (_ffi.Range * len(ranges))(*ranges)
    # host planner (a host-only context plans from the length alone)
    eng = _engine(blocks, titv, seed)
    cid = eng.add_contig(np.zeros(L, dtype=np.uint8))
    eng.plan_contig(cid, list(ranges))
    hrecs, hpool = eng.fetch_records(cid)
    hempty = eng.plan_was_empty(cid)
    hstates = [eng.get_mt_state(0), eng.get_mt_state(1)]
    eng.close()
    # the engine, emulated
    eng = _engine(blocks, titv, seed)
    lib = _dbg(eng.lib)
    recs = np.zeros(sum(r.k for r in ranges) + 8, dtype=_ffi.RECORD_DTYPE)
    pool = np.zeros(int(sum(r.k * max(1, r.max_len[2]) for r in ranges)) + 8, dtype=np.uint8)
    n_recs, pool_len, empty = C.c_uint64(), C.c_uint64(), C.c_int()
    rc = lib.msim_dbg_multimix_plan(eng.h, L, arr, len(ranges), C.c_void_p(recs.ctypes.data), len(recs),
                                    C.c_void_p(pool.ctypes.data), len(pool), C.byref(n_recs), C.byref(pool_len), C.byref(empty))
    if expect_unsupported:
        assert rc == _ffi.ERR_UNSUPPORTED
        eng.close()
        return None
    assert rc == 0, eng.lib.msim_last_error(eng.h)
    states = [eng.get_mt_state(0), eng.get_mt_state(1)]
    eng.close()
    got = recs[:n_recs.value]
    assert len(got) == len(hrecs)
    for f in ("pos", "type", "stop", "aux", "extra"):
        assert np.array_equal(got[f], hrecs[f]), f
    assert np.array_equal(pool[:pool_len.value], hpool)
    assert bool(empty.value) == bool(hempty)
    for (hm, hp), (gm, gp) in zip(hstates, states):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)
    return hrecs


C3_CHANCES = {1: 0.005, 2: 0.001, 3: 0.001, 4: 0.0005, 5: 0.0005}
C3_LENS = {2: (1, 50), 3: (1, 50), 4: (50, 500), 5: (50, 500)}


def _gene_block_layout(L, rs, n_blocks):
    cuts = np.sort(rs.choice(np.arange(1, L - 1), size=2 * n_blocks, replace=False))
    out, at = [], 0
    for a, b in zip(cuts[0::2], cuts[1::2]):
        if a - 1 > at:
            out.append((at, int(a) - 1))
        at = int(b) + 1
    if at < L - 1:
        out.append((at, L - 1))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gene_blocks_with_sv_std_line(seed):
    """The mainstream RMT shape: blocked genes, std line `sn in de du iv` in the gaps (one settings object)."""
    rs = np.random.RandomState(seed)
    L = 3_000_000
    ranges = [_range(s, e, 0.008, C3_CHANCES, C3_LENS) for s, e in _gene_block_layout(L, rs, 150)]
    ranges = [r for r in ranges if r.k]
    recs = _both(L, ranges, None, 2.0, (seed, seed + 5))
    assert len(set(recs["type"].tolist())) == 5


def test_adjacent_ranges_spans_cross_borders():
    """Touching ranges with long DE / DU / IV: a span of range i swallows candidates of range i+1 (blocked range reset per
    range, mutator.py:184; the walk of __mutate_sequence skips them, mutator.py:376,386,398) -- also across a whole tiny
    range into the one after it."""
    L = 400_000
    ranges, at = [], 0
    rs = np.random.RandomState(4)
    while at < L - 10:
        length = int(rs.choice([60, 300, 2_000, 9_000]))
        e = min(L - 1, at + length - 1)
        ranges.append(_range(at, e, 0.03, {1: 0.4, 3: 0.25, 4: 0.2, 5: 0.1, 2: 0.05},
                             {3: (200, 900), 4: (100, 700), 5: (50, 800), 2: (1, 9)}))
        at = e + 1
    ranges = [r for r in ranges if r.k]
    hrecs = _both(L, ranges, None, 1.0, (9, 9))
    # the situation really occurs: some record starts inside the span of the record before it in the raw candidate walk
    assert len(hrecs) > 1000


def test_hot_cold_ranges_with_own_lengths_and_token_order():
    """Ranges with different settings: own chances in RMT token order, own length bounds (-> several randint classes
    in one table), a pool-path hot spot, SNP-only ranges in between, d = 2."""
    L = 1_200_000
    blocks = {t: 2 for t in range(1, 8)}
    blocks[3] = 5
    ranges = [
        _range(0, 199_999, 0.01, {1: 1.0}, {}),
        _range(200_000, 200_999, 0.2, {4: 0.1, 1: 0.8, 2: 0.1}, {4: (5, 8), 2: (1, 4)}, order=[4, 1, 2]),     # pool path
        _range(201_000, 499_999, 0.008, C3_CHANCES, C3_LENS),
        _range(500_000, 500_040, 0.1, {1: 0.5, 3: 0.5}, {3: (1, 50)}),                                       # k = 4
        _range(500_100, 899_999, 0.02, {1: 0.3, 3: 0.3, 5: 0.4}, {3: (1, 50), 5: (50, 500)}, order=[5, 3, 1]),
        _range(900_000, 1_199_999, 0.004, {2: 1.0}, {2: (3, 3)}),                                            # width-1 randint
    ]
    assert (ranges[1].stop - (ranges[1].k - 1) * 2) - ranges[1].start <= ranges[1].setsize
    _both(L, ranges, blocks, 0.5, (21, 22))


@pytest.mark.parametrize("sn_block", [2, 7])
def test_snp_block_above_sampling_distance(sn_block):
    """sn_block > min(block): a kept SNP blocks its successors (mutator.py:204-206), so every candidate is on the
    chain -- SNP-only ranges included."""
    L = 800_000
    rs = np.random.RandomState(sn_block)
    ranges = []
    for s, e in _gene_block_layout(L, rs, 40):
        kind = rs.randint(0, 3)
        if kind == 0:
            ranges.append(_range(s, e, 0.05, {1: 1.0}, {}))
        elif kind == 1:
            ranges.append(_range(s, e, 0.3, {1: 0.9, 2: 0.1}, {2: (1, 4)}))
        else:
            ranges.append(_range(s, e, 0.02, {1: 0.2, 3: 0.4, 5: 0.2, 4: 0.2}, {3: (3, 60), 5: (5, 90), 4: (4, 80)}))
    ranges = [r for r in ranges if r.k and (r.stop - (r.k - 1)) - r.start >= r.k]
    _both(L, ranges, {1: sn_block}, 1.7, (3, 4))


def test_iv_drop_and_clamps_at_the_contig_end():
    L = 300_000
    ranges = [_range(0, 149_999, 0.01, {1: 0.5, 5: 0.5}, {5: (2, 100)}),
              _range(150_000, L - 1, 0.01, {1: 0.2, 5: 0.3, 3: 0.25, 4: 0.25}, {5: (2, 60_000), 3: (1, 90_000), 4: (1, 90_000)})]
    _both(L, ranges, None, 1.0, (5, 6))


@pytest.mark.parametrize("case", range(10))
def test_random_layouts(case):
    rs = np.random.RandomState(500 + case)
    L = int(rs.randint(200_000, 1_500_000))
    d = int(rs.choice([1, 1, 2, 4]))
    blocks = {t: d + int(rs.choice([0, 0, 1, 5, 40])) for t in range(2, 8)}
    blocks[1] = d if rs.rand() < 0.7 else d + int(rs.randint(1, 4))
    blocks[int(rs.choice([2, 3, 4, 5]))] = d
    n_sets = int(rs.randint(1, 5))
    sets = []
    widths = [(1, 1 + int(rs.choice([0, 3, 49, 450]))) for _ in range(2)]      # at most 4 classes in total
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
        order = [int(x) for x in rs.permutation(types)]
        sets.append((chances, lens, order, float(rs.choice([0.002, 0.01, 0.03, 0.1]))))
    ranges, at = [], int(rs.randint(0, 500))
    while at < L - 50:
        length = int(min(L - at, rs.choice([8, 30, 200, 2_000, 20_000, 150_000])))
        chances, lens, order, rate = sets[int(rs.randint(0, n_sets))]
        r = _range(at, at + length - 1, rate, chances, lens, order)
        n = (r.stop - (r.k - 1) * d) - r.start
        if r.k > 0 and n >= r.k:
            ranges.append(r)
        at += length + int(rs.choice([0, 0, 1, 50, 3_000]))
    assert ranges
    _both(L, ranges, blocks, float(rs.choice([0.0, 1.0, 2.0])), (case + 1, case + 11))


def test_refuses_what_belongs_to_the_host_planner():
    L = 200_000
    tl = _range(0, L - 1, 0.01, {1: 0.5, 6: 0.5}, {6: (1, 20)}, order=[1, 6])
    _both(L, [tl], None, 1.0, (1, 1))                                            # (translocations: taken since round 3)
    a, b = _range(0, 99_999, 0.01, C3_CHANCES, C3_LENS), _range(50_000, 150_000, 0.01, C3_CHANCES, C3_LENS)
    _both(L, [a, b], None, 1.0, (1, 1), expect_unsupported=True)                 # overlapping: dict semantics
    five = [_range(i * 30_000, i * 30_000 + 29_999, 0.01, {1: 0.5, 3: 0.5}, {3: (1, 10 + 7 * i)}) for i in range(5)]
    _both(L, five, None, 1.0, (1, 1))                                            # (five randint classes: wide tables, late round 3)
    nine = [_range(i * 20_000, i * 20_000 + 19_999, 0.02, {1: 0.5, 3: 0.5}, {3: (1, 10 + 7 * i)}) for i in range(9)]
    _both(L, nine, None, 1.0, (1, 1), expect_unsupported=True)                   # nine


# ---------------------------------------------------------------------- translocations (mutator.py:267-316)
TL_CHANCES = {1: 0.01, 2: 0.01, 3: 0.01, 5: 0.01, 4: 0.01, 6: 0.005, 7: 0.005}     # README: -sn -in -de -du -iv -tl 0.01 each
TL_LENS = {2: (10, 100), 3: (1, 2), 4: (1, 2), 6: (1, 2)}


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_translocations_readme_flags(seed):
    """The reference's own benchmark flags (README "Performance"): TL spans and TLI sites linked behind the last range --
    __fix_tl_amount's random deletions, shuffle, one coin per pair -- in ARGS order with the TLI chance appended."""
    L = 1_500_000
    r = _range(0, L - 1, 0.06, TL_CHANCES, TL_LENS, order=ARGS_ORDER)
    recs = _both(L, [r], None, 1.0, (seed, seed + 3))
    assert {6, 7} <= set(recs["type"].tolist())


def test_translocations_across_ranges_and_tombstones():
    """Several ranges with TL settings of their own (token order, longer spans, own blocks), SNP-only ranges between them,
    far more TLI sites than TL spans in one range and the opposite in another (many deletions by __fix_tl_amount), spans
    crossing range borders, d = 2."""
    L = 900_000
    blocks = {t: 2 for t in range(1, 8)}
    blocks[6] = 9
    ranges = [
        _range(0, 199_999, 0.02, {1: 0.5, 6: 0.1, 7: 0.4}, {6: (5, 60)}, order=[7, 1, 6]),
        _range(200_000, 399_999, 0.01, {1: 1.0}, {}),
        _range(400_000, 599_999, 0.02, {1: 0.3, 6: 0.45, 7: 0.05, 3: 0.2}, {6: (200, 900), 3: (200, 900)}, order=[6, 3, 7, 1]),
        _range(600_000, 600_300, 0.1, {6: 0.5, 7: 0.5}, {6: (5, 60)}, order=[6, 7]),
        _range(600_301, L - 1, 0.015, {1: 0.4, 2: 0.2, 6: 0.2, 7: 0.2}, {2: (1, 9), 6: (5, 60)}, order=ARGS_ORDER),
    ]
    recs = _both(L, ranges, blocks, 2.0, (13, 14))
    assert (recs["type"] == 7).sum() > 100 and (recs["type"] == 6).sum() > 100


def test_translocation_sites_without_spans_stay_unlinked():
    """tls empty: __link_tls is not called (mutator.py:130), every TLI keeps start = pos, stop = 0."""
    L = 300_000
    recs = _both(L, [_range(0, L - 1, 0.01, {1: 0.5, 7: 0.5}, {}, order=[1, 7])], None, 1.0, (2, 2))
    t = recs[recs["type"] == 7]
    assert len(t) > 500 and np.all(t["stop"] == 0) and np.array_equal(t["extra"], t["pos"])


@pytest.mark.parametrize("case", range(6))
def test_random_layouts_with_translocations(case):
    rs = np.random.RandomState(900 + case)
    L = int(rs.randint(200_000, 1_000_000))
    d = int(rs.choice([1, 1, 2]))
    blocks = {t: d + int(rs.choice([0, 0, 1, 5])) for t in range(2, 8)}
    blocks[1] = d if rs.rand() < 0.7 else d + 1
    blocks[int(rs.choice([2, 3, 4, 5]))] = d
    width = (1, 1 + int(rs.choice([1, 9, 60])))
    sets = []
    for _ in range(int(rs.randint(1, 4))):
        types = [1] + [t for t in (2, 3, 4, 5, 6, 7) if rs.rand() < 0.6]
        chances = {t: float(rs.uniform(0.05, 1.0)) for t in types}
        lens = {t: (width[0] + int(rs.randint(0, 20)),) * 1 + (0,) for t in (2, 3, 4, 6)}
        lens = {t: (a, a + width[1] - width[0]) for t, (a, _) in lens.items()}
        lens[5] = (2, 2 + width[1] - width[0])
        sets.append((chances, lens, [int(x) for x in rs.permutation(types)], float(rs.choice([0.005, 0.02, 0.06]))))
    ranges, at = [], 0
    while at < L - 50:
        length = int(min(L - at, rs.choice([30, 2_000, 20_000, 150_000])))
        chances, lens, order, rate = sets[int(rs.randint(0, len(sets)))]
        r = _range(at, at + length - 1, rate, chances, lens, order)
        if r.k > 0 and (r.stop - (r.k - 1) * d) - r.start >= r.k:
            ranges.append(r)
        at += length + int(rs.choice([0, 0, 1, 500]))
    _both(L, ranges, blocks, 1.0, (case + 3, case + 30))


def test_five_to_eight_randint_classes_take_the_wide_tables():
    """Five SV types with five different length widths (and more, over several settings objects of one contig): beyond four
    classes the accept tables hold 8 slots per word position and the entries 9 bits of slot increment above 23 of value."""
    L = 600_000
    five = {3: (1, 9), 2: (1, 30), 5: (2, 120), 4: (5, 700), 6: (3, 50)}             # widths 9, 30, 119, 696, 48
    chances = {1: 0.3, 3: 0.15, 2: 0.15, 5: 0.1, 4: 0.1, 6: 0.12, 7: 0.08}
    ranges = [_range(0, 299_999, 0.02, chances, five), _range(300_000, L - 1, 0.03, chances, five)]
    _both(L, ranges, {3: 3, 6: 4}, 2.0, (5, 6))
    other = {3: (2, 70), 2: (1, 5), 5: (2, 14), 4: (5, 700), 6: (3, 50)}              # + 69, 5, 13: eight classes on the contig
    ranges = [_range(0, 199_999, 0.02, chances, five), _range(200_000, 399_999, 0.03, chances, other),
              _range(400_000, L - 1, 0.01, {1: 0.5, 2: 0.5}, {2: (1, 30)})]
    _both(L, ranges, {5: 2}, 1.0, (7, 8))
    ninth = {3: (1, 9), 2: (1, 30), 5: (2, 120), 4: (5, 700), 6: (1, 1000)}           # a ninth width: the host planner's
    ranges = [_range(0, 199_999, 0.02, chances, five), _range(200_000, 399_999, 0.03, chances, other),
              _range(400_000, L - 1, 0.01, chances, ninth)]
    _both(L, ranges, {}, 1.0, (9, 10), expect_unsupported=True)
    wide = {3: (1, 9), 2: (1, 30), 5: (2, 120), 4: (5, 700), 6: (1, 1 << 23)}          # 2^23 values do not fit beside 9 bits
    _both(L, [_range(0, L - 1, 0.02, chances, wide)], {}, 1.0, (11, 12), expect_unsupported=True)
