"""C-ABI checks that need no GPU: the library loads, exports every symbol include/msim.h declares,
fails loudly without a device, and its host planner + VCF renderer reproduce the reference goldens
(through a host-only context, which can plan and render but cannot produce a sequence)."""
from __future__ import annotations

import random
import re
from pathlib import Path

import numpy as np
import pytest

from helpers import CASES, ROOT, all_case_names, case_meta, sha256
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from pipeline import plan_only_vcf


def test_header_symbols_exported_and_bound():
    header = (ROOT / "include" / "msim.h").read_text()
    declared = set(re.findall(r"\b(msim_[a-z0-9_]+)\s*\(", header))
    lib = _ffi.load()
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    assert lib.msim_abi_version() == 8


def test_struct_layouts_match_header():
    assert _ffi.C.sizeof(_ffi.Record) == 16 and _ffi.RECORD_DTYPE.itemsize == 16
    assert _ffi.C.sizeof(_ffi.Range) == 8 * 4 + 4 + 4 * 8 + 4 + 8 * 8 * 3
    assert _ffi.C.sizeof(_ffi.Params) == 72
    assert _ffi.C.sizeof(_ffi.Timing) == 5 * 8 + 15 * 8 + 5 * 8


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse, not compute on the CPU."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from mutation_simulator_amd import _ffi\n"
            "import ctypes\n"
            "try:\n    _ffi.Engine(0); print('CREATED')\nexcept _ffi.MsimError as e:\n    print('REFUSED', e)\n"
            % str(ROOT / "mutation-simulator_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "", "PATH": "/usr/bin:/bin"})
    assert "REFUSED" in out.stdout or "CREATED" in out.stdout
    if "CREATED" in out.stdout:
        pytest.skip("a GPU is visible even with empty visibility masks")
    eng = _ffi.Engine(device=-1)
    cid = eng.add_contig(np.frombuffer(b"ACGT" * 10, dtype=np.uint8))
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    eng.set_params(p)
    eng.plan_contig(cid, [])
    with pytest.raises(_ffi.MsimError, match="needs the GPU"):
        eng.apply_contig(cid)
    with pytest.raises(_ffi.MsimError, match="needs the GPU"):
        eng.add_contig_synthetic(100, 1)


def test_mt_state_roundtrip_with_python_generators():
    random.seed(1234)
    np.random.seed(4321)
    [random.random() for _ in range(700)]
    np.random.random_sample(333)
    eng = _ffi.Engine(device=-1)
    mm.export_python_streams(eng)
    want_py = random.getrandbits(32)
    want_np = int(np.random.randint(0, 4294967296, dtype=np.uint32))
    random.seed(0)
    np.random.seed(0)
    mm.import_python_streams(eng)
    assert random.getrandbits(32) == want_py
    assert int(np.random.randint(0, 4294967296, dtype=np.uint32)) == want_np


def test_threshold_arithmetic_matches_float_compare():
    """Integer thresholds == the reference's double comparisons, probed at the boundaries."""
    rng = np.random.RandomState(5)
    for _ in range(200):
        p = rng.random_sample(rng.randint(1, 8))
        p[rng.randint(0, len(p))] = 0.0 if len(p) > 1 else p[0]
        p = p / p.sum()
        cdf = np.cumsum(p.astype(np.float64))
        cdf /= cdf[-1]
        thr = [mm._ceil_scaled(float(c)) for c in cdf]
        probes = set()
        for t in thr:
            probes.update(m for m in (t - 1, t, t + 1) if 0 <= m < (1 << 53))
        probes.update(int(x) for x in rng.randint(0, 2**53, 20, dtype=np.int64))
        for m in probes:
            u = m / 9007199254740992.0
            assert sum(1 for t in thr if t <= m) == int(np.searchsorted(cdf, u, side="right"))
    for titv in [0.0, 0.5, 1.0, 2.0, 2.5, 1e9, 1e-9, 3.0, float("inf")]:
        p_ti = titv * (1 / (titv + 1))
        class S:  # noqa: D401
            mut_block = {}
        S.titv = titv
        lim = mm.params_descriptor(S).ti_lim
        for m in {0, 1, lim - 1, lim, lim + 1, (1 << 53) - 1} - {-1}:
            if 0 <= m < (1 << 53):
                assert (m < lim) == (m / 9007199254740992.0 <= p_ti), (titv, m)


def test_setsize_matches_cpython():
    for k in [0, 1, 5, 6, 7, 21, 22, 85, 86, 211, 341, 342, 1365, 1366, 10000, 2_480_000, 87381, 87382]:
        want = 21 + (4 ** __import__("math").ceil(__import__("math").log(k * 3, 4)) if k > 5 else 0)
        assert mm.sample_setsize(k) == want


RUNNABLE = [n for n in all_case_names() if case_meta(n).get("sim") is not None]


@pytest.mark.parametrize("name", RUNNABLE)
def test_host_planner_vcf_matches_reference(name, tmp_path):
    """settings -> msim_plan_contig (host-only ctx) -> msim_render_vcf == the reference's VCF."""
    meta = case_meta(name)
    if meta["exception"] is not None and meta["exception"]["type"] == "ValueError":
        with pytest.raises(ValueError) as ei:
            plan_only_vcf(meta, tmp_path)
        assert str(ei.value) == meta["exception"]["message"]
        return
    if meta["exception"] is not None:
        pytest.skip("KeyError surfaces in APPLY (GPU test)")
    if "vcf_len" not in meta:
        pytest.skip("`it` mode: no mutation pass (tests/test_it_host.py)")
    vcf, empty, eng = plan_only_vcf(meta, tmp_path)
    assert len(vcf) == meta["vcf_len"] and sha256(vcf) == meta["vcf_sha256"]
    if meta["store"] == "full":
        assert vcf == (CASES / name / "expected_ms.vcf").read_bytes()
    warned = [int(l.split("sequence ")[1].split(" ")[0]) - 1
              for l in meta["stderr"].splitlines() if "No mutations could be generated" in l]
    assert empty == warned
    # both Python generators stand where the reference left them
    if "it_fasta_len" not in meta:              # (with an IT pass behind it the golden's CPython position is the IT pass's)
        assert [random.getrandbits(32) for _ in range(4)] == meta["py_next_words_after"]
    assert [int(x) for x in np.random.randint(0, 4294967296, size=4, dtype=np.uint32)] == \
        meta["np_next_words_after"]


# ---------------------------------------------------------------------- host chains of the device engines
# sample_ranges_host / chain_boundary_host (plan_host.cpp) run on the host but read words the DEVICE generated,
# so the GPU tests exercise them only on a GPU box.  The debug exports let this tier pin them directly against
# CPython's own random.sample / random.randint, word for word.
def _dbg(lib):
    import ctypes as C
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    lib.msim_dbg_cut_ranges.restype = C.c_int
    lib.msim_dbg_cut_ranges.argtypes = [vp, C.POINTER(_ffi.Range), C.c_int, vp, C.c_uint64, vp, vp, u64p, u64p]
    lib.msim_dbg_sample_ranges.restype = C.c_int
    lib.msim_dbg_sample_ranges.argtypes = [vp, C.POINTER(_ffi.Range), C.c_int, vp, C.c_uint64, vp, u64p]
    lib.msim_dbg_chain_boundary_tables.restype = C.c_int
    lib.msim_dbg_chain_boundary_tables.argtypes = [vp, C.POINTER(_ffi.Range), C.c_uint64, vp, vp, C.c_uint64, vp, C.c_uint64,
                                                   vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    lib.msim_dbg_chain_boundary.restype = C.c_int
    lib.msim_dbg_chain_boundary.argtypes = [vp, C.POINTER(_ffi.Range), C.c_uint64, vp, vp, C.c_uint64, vp, C.c_uint64,
                                            vp, u64p, u64p, C.POINTER(C.c_int64)]
    return lib


def _host_engine(blocks=None):
    eng = _ffi.Engine(device=-1)
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    for t, v in (blocks or {}).items():
        p.block[t] = v
    p.ti_lim = (1 << 52) + 1
    eng.set_params(p)
    return eng


def _snp_only_range(start, stop, k):
    r = _ffi.Range()
    r.start, r.stop, r.k = start, stop, k
    r.setsize = mm.sample_setsize(k)
    r.n_types = 1
    r.types[0] = 1
    r.cdf_thr[0] = 1 << 53
    return r


@pytest.mark.parametrize("seed,d", [(1, 1), (2, 1), (3, 3), (4, 2)])
def test_host_range_sampler_equals_cpython_sample(seed, d):
    """Chains of random.sample() calls over set-path and pool-path ranges of very different sizes, minimum
    distance d: positions and the exact number of words consumed equal CPython's."""
    import ctypes as C
    rs = np.random.RandomState(seed)
    ranges, at = [], 0
    for i in range(300):
        length = int(rs.choice([12, 40, 300, 1000, 5000, 60_000, 700_000]))
        rate = float(rs.choice([0.001, 0.01, 0.05, 0.2, 0.3]))
        k = int(length * rate)
        n = (at + length - 1 - (k - 1) * d) - at
        if k > 0 and n >= k:
            ranges.append(_snp_only_range(at, at + length - 1, k))
        at += length + int(rs.randint(1, 500))
    assert any((r.stop - (r.k - 1) * d) - r.start <= r.setsize for r in ranges)      # pool path present
    assert any((r.stop - (r.k - 1) * d) - r.start > 4096 * 64 for r in ranges)       # large-bitmap path present
    K = sum(r.k for r in ranges)
    ref = random.Random(seed)
    clone = random.Random(seed)
    words = np.array([clone.getrandbits(32) for _ in range(3 * K + 100_000)], dtype=np.uint32)
    want = []
    for r in ranges:
        n = (r.stop - (r.k - 1) * d) - r.start
        vals = sorted(ref.sample(range(n), r.k))                                       # util.py:104-109
        want += [r.start + v + d * i for i, v in enumerate(vals)]
    nxt = ref.getrandbits(32)
    eng = _host_engine({t: d for t in range(1, 8)})
    lib = _dbg(eng.lib)
    got = np.zeros(K + 8, dtype=np.uint32)
    used = C.c_uint64()
    arr = (_ffi.Range * len(ranges))(*ranges)
    rc = lib.msim_dbg_sample_ranges(eng.h, arr, len(ranges), C.c_void_p(words.ctypes.data), len(words),
                                    C.c_void_p(got.ctypes.data), C.byref(used))
    assert rc == 0
    assert got[:K].tolist() == want
    assert int(words[used.value]) == nxt                    # exactly as many words consumed as CPython drew
    # a window that is too short is reported, never silently truncated
    rc = lib.msim_dbg_sample_ranges(eng.h, arr, len(ranges), C.c_void_p(words.ctypes.data), used.value - 1,
                                    C.c_void_p(got.ctypes.data), C.byref(used))
    assert rc != 0
    eng.close()


@pytest.mark.parametrize("seed,d", [(1, 1), (2, 1), (3, 3), (4, 2)])
def test_host_range_cuts_reproduce_cpython_sample(seed, d):
    """The host-cut engine's host half: only WHERE each random.sample() starts in the word stream (plus the pool-path
    draws).  Replaying the device half here -- accepted draws of [cut[i], cut[i+1]) as a set, sorted, + d * rank
    (k_interval_bits / k_walk_expand) -- must give CPython's positions, and the cuts CPython's word consumption.
    Range sizes cover all three bitmap regimes of cut_ranges_host: L1 (plain inserts, memset), beyond L1 (prefetched
    inserts) and beyond 256 KB (cleared draw by draw)."""
    import ctypes as C
    rs = np.random.RandomState(seed)
    ranges, at = [], 0
    for i in range(300):
        length = int(rs.choice([12, 40, 300, 1000, 5000, 60_000, 700_000]))
        if i in (40, 170):
            length = 3_000_000                                                         # bitmap of 375 KB
        rate = float(rs.choice([0.001, 0.01, 0.05, 0.2, 0.3]))
        if length == 3_000_000:
            rate = 0.002
        k = int(length * rate)
        n = (at + length - 1 - (k - 1) * d) - at
        if k > 0 and n >= k:
            ranges.append(_snp_only_range(at, at + length - 1, k))
        at += length + int(rs.randint(1, 500))
    K = sum(r.k for r in ranges)
    ref = random.Random(seed)
    clone = random.Random(seed)
    words = np.array([clone.getrandbits(32) for _ in range(3 * K + 100_000)], dtype=np.uint32)
    want = []
    for r in ranges:
        n = (r.stop - (r.k - 1) * d) - r.start
        vals = sorted(ref.sample(range(n), r.k))                                       # util.py:104-109
        want.append([r.start + v + d * i for i, v in enumerate(vals)])
    nxt = ref.getrandbits(32)
    eng = _host_engine({t: d for t in range(1, 8)})
    lib = _dbg(eng.lib)
    is_pool = [(r.stop - (r.k - 1) * d) - r.start <= r.setsize for r in ranges]
    assert any(is_pool) and not all(is_pool)
    cut = np.zeros(len(ranges) + 1, dtype=np.uint32)
    pool = np.zeros(sum(r.k for r, p in zip(ranges, is_pool) if p) + 8, dtype=np.uint32)
    used, n_pool = C.c_uint64(), C.c_uint64()
    arr = (_ffi.Range * len(ranges))(*ranges)
    rc = lib.msim_dbg_cut_ranges(eng.h, arr, len(ranges), C.c_void_p(words.ctypes.data), len(words),
                                 C.c_void_p(cut.ctypes.data), C.c_void_p(pool.ctypes.data), C.byref(n_pool), C.byref(used))
    assert rc == 0
    assert int(words[used.value]) == nxt and int(cut[-1]) == used.value
    assert np.all(np.diff(cut.astype(np.int64)) > 0)
    pool_at = 0
    for i, r in enumerate(ranges):
        n = (r.stop - (r.k - 1) * d) - r.start
        if is_pool[i]:
            vals = np.sort(pool[pool_at:pool_at + r.k].astype(np.int64))
            pool_at += r.k
        else:
            w = words[cut[i]:cut[i + 1]] >> np.uint32(32 - int(n).bit_length())
            vals = np.unique(w[w < n]).astype(np.int64) + r.start
        assert len(vals) == r.k
        assert (vals + d * np.arange(r.k)).tolist() == want[i]
    assert pool_at == n_pool.value
    rc = lib.msim_dbg_cut_ranges(eng.h, arr, len(ranges), C.c_void_p(words.ctypes.data), used.value - 1,
                                 C.c_void_p(cut.ctypes.data), C.c_void_p(pool.ctypes.data), C.byref(n_pool), C.byref(used))
    assert rc != 0
    eng.close()


def test_plan_mode_can_be_changed_on_a_live_context():
    """msim_set_plan_mode: AUTO / HOST on any context, GPU only where a device exists, nonsense is refused."""
    eng = _host_engine()
    eng.set_plan_mode(_ffi.PLAN_HOST)
    eng.set_plan_mode(_ffi.PLAN_AUTO)
    for bad in (_ffi.PLAN_GPU, _ffi.PLAN_HOST | _ffi.PLAN_GPU, 8):
        with pytest.raises(_ffi.MsimError):
            eng.set_plan_mode(bad)
    eng.close()


def _sv_range(L, n):
    """A range whose type draw can produce IN, DE, DU and IV (the boundary walk ignores the lengths of types it cannot)."""
    r = _ffi.Range()
    r.start, r.stop, r.k = 0, L - 1, n
    r.n_types = 4
    for j, t in enumerate((2, 3, 4, 5)):
        r.types[j] = t
        r.cdf_thr[j] = (j + 1) << 51
    return r


def test_host_boundary_tables_report_short_window_and_foreign_type():
    """The table walk fails like the plain one: a window that ends before the chain does is an error, not a short
    read, and so is a candidate type that does not belong to the boundary pass."""
    import ctypes as C
    rs = np.random.RandomState(9)
    L, n = 100_000, 2_000
    pos = np.sort(rs.choice(np.arange(0, L - 1), size=n, replace=False)).astype(np.uint32)
    types = rs.choice([2, 3, 4, 5], size=n).astype(np.uint8)
    r = _sv_range(L, n)
    for t in (2, 3, 4, 5):
        r.min_len[t], r.max_len[t] = 1, 5
    words = rs.randint(0, 2**32, size=4 * n, dtype=np.uint64).astype(np.uint32)
    eng = _host_engine({t: 1 for t in range(1, 8)})
    lib = _dbg(eng.lib)
    stop = np.zeros(n, dtype=np.uint32)
    used, nk, dl = C.c_uint64(), C.c_uint64(), C.c_int64()

    def run(fn, ty, nw):
        return fn(eng.h, C.byref(r), L, C.c_void_p(pos.ctypes.data), C.c_void_p(ty.ctypes.data), n,
                  C.c_void_p(words.ctypes.data), nw, C.c_void_p(stop.ctypes.data), C.byref(used), C.byref(nk), C.byref(dl))
    for fn in (lib.msim_dbg_chain_boundary, lib.msim_dbg_chain_boundary_tables):
        assert run(fn, types, len(words)) == 0
        full = used.value
        assert run(fn, types, full) == 0 and used.value == full           # the window may end exactly where the chain does
        assert run(fn, types, full - 1) != 0
        bad = types.copy()
        bad[0] = 1                                                         # an SNP is not a boundary candidate
        assert run(fn, bad, len(words)) != 0
    r.max_len[5] = 1 << 25                                                 # width beyond a table entry's value field
    assert run(lib.msim_dbg_chain_boundary_tables, types, len(words)) != 0
    # ... which only matters for a type the range can draw: with IV out of the type table the tables are back
    r.n_types = 3
    no_iv = np.where(types == 5, 4, types).astype(np.uint8)
    assert run(lib.msim_dbg_chain_boundary, no_iv, len(words)) == 0
    want = (stop.copy(), used.value, nk.value, dl.value)
    assert run(lib.msim_dbg_chain_boundary_tables, no_iv, len(words)) == 0
    assert (stop.tolist(), used.value, nk.value, dl.value) == (want[0].tolist(),) + want[1:]
    eng.close()


@pytest.mark.parametrize("walker", ["words", "tables"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_host_boundary_chain_equals_cpython_randint(seed, walker):
    """The boundary pass over non-SNP candidates (mutator.py:184-265) restated with CPython's randint:
    same stops, same drops (blocked / inversion reaching the contig end), same words consumed, same length delta.
    `words`: the plain walk over tempered words; `tables`: the walk over "next accepted draw" tables that the SV-mix
    engine uses (built here on the host, on the device by k_accept_tables)."""
    import ctypes as C
    rs = np.random.RandomState(seed)
    L = 400_000
    n = 20_000
    pos = np.sort(rs.choice(np.arange(0, L - 1), size=n, replace=False)).astype(np.uint32)
    types = rs.choice([2, 3, 4, 5], size=n).astype(np.uint8)                 # IN, DE, DU, IV
    lens = {2: (1, int(rs.randint(1, 40))), 3: (1, int(rs.randint(1, 300))), 4: (2, int(rs.randint(2, 90))),
            5: (2, int(rs.randint(2, 500)))}
    if seed == 5:
        lens[2] = (7, 7)                                                        # width-1 randint burns words until a 0 bit
    block = {1: 1, 2: int(rs.randint(1, 4)), 3: int(rs.randint(1, 30)), 4: 1, 5: int(rs.randint(1, 9)), 6: 1, 7: 1}
    r = _sv_range(L, n)
    for t, (a, b) in lens.items():
        r.min_len[t], r.max_len[t] = a, b
    ref = random.Random(seed)
    clone = random.Random(seed)
    words = np.array([clone.getrandbits(32) for _ in range(4 * n + 1000)], dtype=np.uint32)
    want, blk_hi, delta, kept = [], 0, 0, 0
    for p, t in zip(pos.tolist(), types.tolist()):
        if p < blk_hi:
            want.append(0xFFFFFFFF)
            continue
        if t == 5 and p + lens[5][1] >= L - 1:                                  # mutator.py:240-245
            want.append(0xFFFFFFFF)
            continue
        s = ref.randint(p + lens[t][0] - 1, p + lens[t][1] - 1)
        if t in (3, 4) and s > L - 1:
            s = L - 1
        want.append(s)
        blk_hi = (p if t == 2 else s) + 1 + block[t]
        delta += {2: 1, 3: -1, 4: 1, 5: 0}[t] * (s - p + 1)
        kept += 1
    nxt = ref.getrandbits(32)
    eng = _host_engine(block)
    lib = _dbg(eng.lib)
    stop = np.zeros(n, dtype=np.uint32)
    used, nk, dl = C.c_uint64(), C.c_uint64(), C.c_int64()
    fn = lib.msim_dbg_chain_boundary if walker == "words" else lib.msim_dbg_chain_boundary_tables
    rc = fn(eng.h, C.byref(r), L, C.c_void_p(pos.ctypes.data), C.c_void_p(types.ctypes.data), n,
            C.c_void_p(words.ctypes.data), len(words), C.c_void_p(stop.ctypes.data), C.byref(used), C.byref(nk), C.byref(dl))
    assert rc == 0
    assert stop.tolist() == want
    assert nk.value == kept and dl.value == delta
    assert int(words[used.value]) == nxt
    assert 0xFFFFFFFF in want and kept > 1000
    eng.close()


def test_plan_chain_advances_like_plan_contig_on_a_host_only_context():
    """msim_plan_chain without a GPU: the host planner runs, nothing is kept, both streams stand where msim_plan_contig
    leaves them (the CPU tier of ``--gpus N`` relies on it)."""
    from mutation_simulator_amd.mut_types import MutType

    class S:
        mut_block = {t: 1 for t in MutType}
        titv = 2.0
    r = _ffi.Range()
    r.start, r.stop, r.k = 0, 99_999, 2_000
    r.setsize = mm.sample_setsize(r.k)
    r.n_types = 3
    for j, (t, thr) in enumerate(((1, 0.6), (2, 0.8), (3, 1.0))):
        r.types[j], r.cdf_thr[j] = t, int(thr * (1 << 53))
    r.min_len[2], r.max_len[2], r.min_len[3], r.max_len[3] = 1, 9, 2, 30
    states = []
    for chain in (False, True):
        eng = _ffi.Engine(device=-1)
        eng.seed(3, 9)
        eng.set_params(mm.params_descriptor(S))
        for _ in range(3):
            if chain:
                eng.plan_chain(100_000, [r])
            else:
                eng.plan_contig(eng.add_contig(np.zeros(100_000, np.uint8)), [r])
        states.append((eng.get_mt_state(0), eng.get_mt_state(1), eng.stats()))
        eng.close()
    (a0, a1, sa), (b0, b1, sb) = states
    assert np.array_equal(a0[0], b0[0]) and a0[1] == b0[1] and np.array_equal(a1[0], b1[0]) and a1[1] == b1[1]
    assert sa["py_words"] == sb["py_words"] and sa["contigs_host"] == 3 and sb["contigs_host"] == 0
