"""Anchored windows of the SNP sampler (plan_gpu.hip: enqueue_sample_ahead; plan_kernels.h: k_ahead_count / k_ahead_fringe):
a sample's count / scatter / de-dup off the stream-position chain, on a window anchored at the host's bound of where the sample
starts, the accepted draws in front of the anchor joined in once the exact start is there.

What has to hold: the same records and both streams at the same positions as the sequential host planner (which restates
``mutator.py:144-265`` and ``util.py:104-109``), whatever the uncertainty of the start, the accept ratio, the duplicate rate,
the number of contigs in flight -- and a start outside the interval the host planned for is an error, never a wrong result.
``MSIM_AHEAD=2`` plans every eligible sample this way (the default since round 6; ``MSIM_AHEAD=1``: only on a rank of a sharded
step -- round 5's default; ``MSIM_NO_AHEAD``: never)."""
from __future__ import annotations

import numpy as np
import pytest

from mutation_simulator_amd import _ffi
from test_gpu_sampler import _next_words, _params, _snp_range

pytestmark = pytest.mark.gpu


def _run_then_fetch(flags, contigs, params, seed, chain_only=()):
    """Plan every contig first (nothing synchronises in between: the device knows where a contig starts, the host only bounds
    it), THEN read.  ``chain_only``: contigs walked with msim_plan_chain."""
    eng = _ffi.Engine(0, flags)
    try:
        eng.seed(*seed)
        eng.set_params(params)
        cids = []
        for i, (L, ranges) in enumerate(contigs):
            if i in chain_only:
                eng.plan_chain(L, ranges)
                cids.append(None)
            else:
                cid = eng.add_contig_synthetic(L, 7)
                eng.plan_contig(cid, ranges)
                cids.append(cid)
        out = []
        for cid in cids:
            if cid is None:
                out.append(None)
                continue
            recs, pool = eng.fetch_records(cid)
            out.append((recs.copy(), pool.copy()))
        st = eng.stats()
        states = [eng.get_mt_state(0), eng.get_mt_state(1)]
        return out, states, st
    finally:
        eng.close()


def _same(host, gpu, hs, gs, hst, gst):
    for h, g in zip(host, gpu):
        if g is None:
            continue
        assert h[0].shape == g[0].shape
        assert np.array_equal(h[0].view(np.uint8), g[0].view(np.uint8))
        assert np.array_equal(h[1], g[1])
    assert hst["py_words"] == gst["py_words"] and hst["np_words"] == gst["np_words"]
    for (hm, hp), (gm, gp) in zip(hs, gs):
        assert _next_words(hm, hp, 8) == _next_words(gm, gp, 8)


def _genome(rs, n_contigs, lo=600_000, hi=6_000_000, rate=(0.004, 0.02)):
    contigs = []
    for _ in range(n_contigs):
        L = int(rs.randint(lo, hi))
        k = max(4096, int(L * rs.uniform(*rate)))
        contigs.append((L, [_snp_range(0, L - 1, k)]))
    return contigs


@pytest.mark.parametrize("seed,titv,n_contigs", [(1, 2.0, 12), (2, 0.0, 9), (3, 1e9, 9), (4, 0.5, 20)])
def test_every_sample_ahead_vs_host_planner(monkeypatch, seed, titv, n_contigs):
    monkeypatch.setenv("MSIM_AHEAD", "2")
    rs = np.random.RandomState(seed)
    contigs = _genome(rs, n_contigs)
    params = _params(titv=titv)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (seed, seed + 1))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (seed, seed + 1))
    _same(host, gpu, hs, gs, hst, gst)
    # (not the first, which starts at an exact position, and not those smaller than twice the uncertainty of their start)
    assert n_contigs // 2 <= gst["snp_samples_ahead"] < n_contigs
    assert gst["contigs_snp"] == n_contigs


def test_accept_ratios_duplicates_and_distance_three(monkeypatch):
    """n just above / below a power of two (accept ratio 0.5 / 1.0), k/n = 0.2 (thousands of duplicates: the fringe pass and the
    tail rounds both find some), sampling distance 3, contigs that draw nothing and a two-range contig (on the chain) in between."""
    monkeypatch.setenv("MSIM_AHEAD", "2")
    blocks = {t: 3 for t in ("SN", "IN", "DE", "IV", "DU", "TL", "TLI")}
    contigs = [(3_000_000, [_snp_range(0, 2_999_999, 30_000)]),
               (4_194_304 + 3 * 40_999 + 1, [_snp_range(0, 4_194_304 + 3 * 40_999, 41_000)]),       # n = 2^22 + 1: 23-bit draws
               (4_194_304 + 3 * 40_999 - 1, [_snp_range(0, 4_194_304 + 3 * 40_999 - 2, 41_000)]),   # n = 2^22 - 1
               (900_000, []),
               (2_400_000, [_snp_range(0, 2_399_999, 300_000)]),                                     # k/n = 0.2 after the distance
               (2_500_000, [_snp_range(0, 999_999, 10_000), _snp_range(1_000_000, 2_499_999, 120_000, True)]),
               (1_700_000, [_snp_range(100_000, 1_650_000, 50_000)]),
               (5_000_000, [_snp_range(0, 4_999_999, 200_000)])]
    params = _params(blocks, titv=1.5)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (3, 4))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (3, 4))
    _same(host, gpu, hs, gs, hst, gst)
    assert gst["snp_samples_ahead"] >= 5


def test_more_contigs_in_flight_than_scratch_sets(monkeypatch):
    """70 contigs planned before the first read: every scratch set (32 of them) meets its third user while the device may still
    be at the first one's chain."""
    monkeypatch.setenv("MSIM_AHEAD", "2")
    rs = np.random.RandomState(11)
    contigs = _genome(rs, 70, lo=3_000_000, hi=5_000_000, rate=(0.03, 0.05))
    params = _params(titv=2.0)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (5, 6))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (5, 6))
    _same(host, gpu, hs, gs, hst, gst)
    assert gst["snp_samples_ahead"] >= 45          # (more than the 32 sets)


def test_sharded_rank_takes_it_by_itself(monkeypatch):
    """Policies: by default (round 6) every context plans ahead; MSIM_AHEAD=1 is round 5's -- a context that walks other ranks'
    contigs (msim_plan_chain) plans ahead, one that owns everything does not; MSIM_NO_AHEAD: nobody."""
    rs = np.random.RandomState(21)
    contigs = _genome(rs, 10, lo=1_000_000, hi=4_000_000)
    params = _params(titv=2.0)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (8, 9))
    full, fs, fst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (8, 9))
    _same(host, full, hs, fs, hst, fst)
    assert fst["snp_samples_ahead"] >= len(contigs) // 2
    monkeypatch.setenv("MSIM_NO_AHEAD", "1")
    none, ns, nst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (8, 9))
    _same(host, none, hs, ns, hst, nst)
    assert nst["snp_samples_ahead"] == 0
    monkeypatch.delenv("MSIM_NO_AHEAD")
    monkeypatch.setenv("MSIM_AHEAD", "1")
    full, fs, fst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (8, 9))
    _same(host, full, hs, fs, hst, fst)
    assert fst["snp_samples_ahead"] == 0
    for owned in ((), (0, 4, 8), (1, 2, 3, 9)):
        others = [i for i in range(len(contigs)) if i not in owned]
        gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (8, 9), chain_only=others)
        _same(host, gpu, hs, gs, hst, gst)
        assert gst["snp_samples_ahead"] >= len(contigs) // 2, (owned, gst["snp_samples_ahead"])
        assert gst["contigs_snp"] == len(owned)


@pytest.mark.parametrize("group", [1, 3, 4])
def test_emission_groups_with_every_sample_ahead(monkeypatch, group):
    monkeypatch.setenv("MSIM_AHEAD", "2")
    monkeypatch.setenv("MSIM_EMIT_GROUP", str(group))
    rs = np.random.RandomState(30 + group)
    contigs = _genome(rs, 9)
    params = _params(titv=2.0)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (2, 2))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (2, 2))
    _same(host, gpu, hs, gs, hst, gst)


def test_a_start_outside_the_interval_is_an_error_not_a_result(monkeypatch):
    """MSIM_AHEAD_SIGMA=0 (test hook): the interval is the 256 words of rounding slack around the expected start, against a
    standard deviation of several hundred words per contig -- a start outside it is certain within a few contigs, and the context
    must report the overflow at the next synchronising call.  With the default interval the same genome plans cleanly."""
    monkeypatch.setenv("MSIM_AHEAD", "2")
    rs = np.random.RandomState(41)
    contigs = _genome(rs, 16, lo=2_000_000, hi=5_000_000)
    params = _params(titv=2.0)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (6, 7))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (6, 7))
    _same(host, gpu, hs, gs, hst, gst)
    monkeypatch.setenv("MSIM_AHEAD_SIGMA", "0")
    with pytest.raises(_ffi.MsimError, match="overflowed"):
        _run_then_fetch(_ffi.PLAN_GPU, contigs, params, (6, 7))


def test_sessions_rebase_between_anchored_windows(monkeypatch):
    """A two-chunk jump-table span (MSIM_DBG_JUMP_MAX_CHUNKS): the session ends every few contigs, the estimate of the position
    restarts at the exact one behind each re-base."""
    monkeypatch.setenv("MSIM_AHEAD", "2")
    monkeypatch.setenv("MSIM_DBG_JUMP_MAX_CHUNKS", "4")
    rs = np.random.RandomState(51)
    contigs = _genome(rs, 14, lo=1_000_000, hi=3_000_000, rate=(0.01, 0.02))
    params = _params(titv=2.0)
    host, hs, hst = _run_then_fetch(_ffi.PLAN_HOST, contigs, params, (10, 11))
    gpu, gs, gst = _run_then_fetch(_ffi.PLAN_AUTO, contigs, params, (10, 11))
    _same(host, gpu, hs, gs, hst, gst)
    assert gst["stream_rebases"] >= 2 and gst["snp_samples_ahead"] >= 4


@pytest.mark.parametrize("owned", [None, (0, 3, 6, 9, 12, 15, 18, 21), ()])
def test_full_size_genome_ahead_vs_oracle(monkeypatch, owned):
    """BASELINE configs[1] at 3 Gb through ``bench.one_step`` (24 contigs enqueued, one synchronisation, then the reads) against
    the oracle per contig: every sample ahead on a rank that owns everything (``MSIM_AHEAD=2``), and by the default policy on a
    rank of a sharded step that owns every third contig / none at all (both streams must still end where the oracle's do)."""
    import bench
    from test_gpu_bench_order import _bench_order_vs_oracle
    if owned is None:
        monkeypatch.setenv("MSIM_AHEAD", "2")
    lengths = bench.contig_lengths(3_000_000_000)
    st = _bench_order_vs_oracle("c2", lengths, owned=owned)
    assert st["snp_samples_ahead"] >= 20, st["snp_samples_ahead"]
    assert st["contigs_snp"] == (24 if owned is None else len(owned))
    # the moment model behind the intervals, observed: the worst of the 23 exact starts used this much of the allowed
    # deviation (8 sigma + 256 words = 1000); 5 sigma of 8 would be ~600
    assert 0 < st["snp_ahead_margin_permille"] < 650, st["snp_ahead_margin_permille"]
