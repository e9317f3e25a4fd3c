"""Text on the device: VCF record lines, FASTA line framing and FASTA ingest done
by HIP kernels must be byte-identical to the host restatements (msim_render_vcf / FastaWriter / the host
FASTA parser), which in turn are pinned to the reference's goldens by the CLI tests."""
from __future__ import annotations

import numpy as np
import pytest

from inputs import decorate, random_bases
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from test_gpu_sampler import ARGS_ORDER, _params, _sv_range

pytestmark = pytest.mark.gpu

ALL_SV = {1: 0.3, 2: 0.15, 3: 0.15, 4: 0.1, 5: 0.1, 6: 0.1, 7: 0.1}
LENS = {2: (1, 9), 3: (1, 12), 4: (2, 30), 5: (2, 25), 6: (1, 15)}


def _wrap(seq: np.ndarray, bpl: int) -> bytes:
    """fasta_writer.py:40-58 for one record body: newline after every full line, none after a partial one."""
    raw = seq.tobytes()
    out = bytearray()
    for i in range(0, len(raw), bpl):
        out += raw[i:i + bpl]
        if i + bpl <= len(raw):
            out += b"\n"
    return bytes(out)


def _plan_apply(eng, bases, ranges):
    cid = eng.add_contig(bases)
    eng.plan_contig(cid, ranges)
    eng.apply_contig(cid)
    return cid


@pytest.mark.parametrize("L,rate,seed,deco", [(200_000, 0.02, 1, True), (1_500_000, 0.008, 2, False),
                                               (50_000, 0.05, 3, True)])
def test_vcf_device_equals_host_renderer(L, rate, seed, deco):
    bases = random_bases(L, seed)
    if deco:
        bases = decorate(bases, seed + 1, n_runs=6, iupac=400, lower=0)
        bases[bases == ord("U")] = ord("A")
    eng = _ffi.Engine(0)
    eng.seed(seed, seed + 10)
    eng.set_params(_params(titv=1.0))
    r = _sv_range(0, L - 1, int(L * rate), ALL_SV, LENS)
    try:
        cid = _plan_apply(eng, bases, [r])
    except KeyError:
        pytest.skip("transversion drawn on an ambiguity code: the reference raises KeyError here")
    recs, pool = eng.fetch_records(cid)
    assert {1, 2, 3, 4, 5}.issubset(set(np.unique(recs["type"]).tolist()))
    want = _ffi.render_vcf(recs, pool, bases, "chrT some name")
    got = eng.render_vcf_device(cid, "chrT some name").tobytes()
    assert got == want
    eng.close()


@pytest.mark.parametrize("lens,seed", [({2: (200, 900), 3: (250, 1500), 4: (256, 700), 5: (300, 1100), 6: (260, 2000)}, 21),
                                       ({2: (20, 300), 3: (24, 26), 4: (25, 257), 5: (1, 600), 6: (23, 258)}, 22)])
@pytest.mark.parametrize("plain", [False, True], ids=["stage", "plain"])
def test_vcf_device_long_records_take_the_wave_path(lens, seed, plain, monkeypatch):
    """REF / ALT of hundreds of bases: their aligned interiors are copied source -> text by the whole wave, 16 bytes per lane
    (raw, converted, reverse-complemented), their edges and the short copies go through the wave's LDS stage, which the flush
    shifts by the gaps; the second mix straddles the inline threshold.  plain: the byte-by-byte path of waves the stage scheme
    cannot describe (4 GiB of text in one wave's 64 lines), forced by the test hook."""
    if plain:
        monkeypatch.setenv("MSIM_DBG_VCF_PLAIN", "1")
    L = 2_000_000
    bases = decorate(random_bases(L, seed), seed + 1, n_runs=4, iupac=300, lower=0)
    bases[bases == ord("U")] = ord("A")
    eng = _ffi.Engine(0)
    eng.seed(seed, seed + 10)
    eng.set_params(_params(titv=1.0))
    mix = {2: 0.2, 3: 0.2, 4: 0.2, 5: 0.2, 6: 0.1, 7: 0.1}
    cid = _plan_apply(eng, bases, [_sv_range(0, L - 1, 1500, mix, lens)])
    recs, pool = eng.fetch_records(cid)
    want = _ffi.render_vcf(recs, pool, bases, "chrLong")
    got = eng.render_vcf_device(cid, "chrLong").tobytes()
    assert got == want and len(want) > 300_000
    eng.close()


def test_vcf_device_tiny_contigs_cover_position_zero_and_contig_end():
    """Thousands of 12-40 base contigs under a dense SV + translocation mix: mutations at position 0
    (IN / DE / TLI special cases, mutator.py:346-358, 362-371, 404-413), deletions clamped at the end."""
    rs = np.random.RandomState(11)
    eng = _ffi.Engine(0, _ffi.PLAN_HOST)
    eng.seed(5, 6)
    eng.set_params(_params(titv=0.8))
    seen_pos0 = set()
    total = 0
    for i in range(1500):
        L = int(rs.randint(12, 41))
        bases = random_bases(L, 1000 + i)
        r = _sv_range(0, L - 1, max(1, int(L * 0.25)), ALL_SV, {2: (1, 4), 3: (1, 6), 4: (2, 5), 5: (2, 5), 6: (1, 5)})
        cid = _plan_apply(eng, bases, [r])
        recs, pool = eng.fetch_records(cid)
        for t in recs["type"][recs["pos"] == 0]:
            seen_pos0.add(int(t))
        want = _ffi.render_vcf(recs, pool, bases, f"c{i}")
        got = eng.render_vcf_device(cid, f"c{i}").tobytes()
        assert got == want, (i, recs)
        total += len(want)
        eng.clear()
    assert {2, 3}.issubset(seen_pos0) and total > 100_000
    eng.close()


def test_vcf_device_snp_only_table_from_gpu_sampler():
    L = 3_000_000
    eng = _ffi.Engine(0)
    eng.seed(42, 42)
    eng.set_params(_params(titv=2.0))
    bases = random_bases(L, 9)
    bases[1000:1200] = ord("N")                          # SNPs on N: REF == ALT, line suppressed
    cid = _plan_apply(eng, bases, [_sv_range(0, L - 1, 30_000, {1: 1.0}, {})])
    recs, pool = eng.fetch_records(cid)
    want = _ffi.render_vcf(recs, pool, bases, "chr1")
    assert eng.render_vcf_device(cid, "chr1").tobytes() == want
    assert want.count(b"\n") <= len(recs)
    eng.close()


@pytest.mark.parametrize("name_len", [1, 36, 37, 38, 39, 40, 41, 150, 1500])
def test_vcf_device_snp_lines_at_the_staging_limit(name_len):
    """The write pass formats the 64 lines of a wave into 4 KB of LDS when they fit (k_vcf_lines, VCF_STAGE) and writes them
    one lane each when they do not: names around the limit (64 lines of name + 7 digits + 19 bytes, plus the stretch's phase),
    a name far beyond it, suppressed lines in between -- on an SNP-only table (its own kernel) and on a mix of short records."""
    L = 2_500_000
    name = ("chr" + "x" * 2000)[:name_len]
    bases = random_bases(L, 31 + name_len)
    bases[5000:5400] = ord("N")                          # REF == ALT: suppressed lines inside staged stretches
    for types, lens, k in (({1: 1.0}, {}, 40_000), ({1: 0.6, 2: 0.2, 3: 0.2}, {2: (1, 3), 3: (1, 3)}, 20_000)):
        eng = _ffi.Engine(0)
        eng.seed(7, 8)
        eng.set_params(_params(titv=2.0))
        cid = _plan_apply(eng, bases, [_sv_range(0, L - 1, k, types, lens)])
        recs, pool = eng.fetch_records(cid)
        want = _ffi.render_vcf(recs, pool, bases, name)
        assert eng.render_vcf_device(cid, name).tobytes() == want
        assert len(want) > 15 * len(recs)
        eng.close()


@pytest.mark.parametrize("bpl", [1, 7, 60, 61, 4096])
def test_framed_fetch_equals_fasta_writer_rule(bpl):
    L = 123_457
    eng = _ffi.Engine(0)
    eng.seed(3, 3)
    eng.set_params(_params())
    bases = random_bases(L, 4)
    cid = _plan_apply(eng, bases, [_sv_range(0, L - 1, 1200, ALL_SV, LENS)])
    seq = eng.fetch_sequence(cid)
    assert eng.fetch_sequence_framed(cid, bpl).tobytes() == _wrap(seq, bpl)
    # exact multiple of the line width: the last line is full and keeps its newline
    cid2 = _plan_apply(eng, random_bases(bpl * 5, 1), [])
    assert eng.fetch_sequence_framed(cid2, bpl).tobytes() == _wrap(eng.fetch_sequence(cid2), bpl)
    eng.close()


def test_framed_fetch_through_the_real_writer(tmp_path):
    """Device-framed bodies written with FastaWriter.write_framed give the same file as write_array."""
    import mutation_simulator_amd as msa
    eng = _ffi.Engine(0)
    eng.seed(1, 2)
    eng.set_params(_params())
    a, b = msa.FastaWriter(tmp_path / "a.fa"), msa.FastaWriter(tmp_path / "b.fa")
    for i, (L, bpl) in enumerate([(1000, 60), (60, 60), (61, 60), (5, 70), (0, 60), (777, 11)]):
        bases = random_bases(L, 20 + i)
        cid = _plan_apply(eng, bases, [])
        for w in (a, b):
            w.set_bpl(bpl)
            w.write_header(f"rec{i} x")
        a.write_array(eng.fetch_sequence(cid))
        b.write_framed(eng.fetch_sequence_framed(cid, bpl), L)
    a.close()
    b.close()
    assert (tmp_path / "a.fa").read_bytes() == (tmp_path / "b.fa").read_bytes()
    eng.close()


@pytest.mark.parametrize("lenc,eol", [(60, b"\n"), (60, b"\r\n"), (1, b"\n"), (113, b"\n")])
def test_ingest_from_text_equals_host_parse(lenc, eol):
    L = 250_003
    bases = decorate(random_bases(L, 8), 9, lower=12)                      # lower-case stretches included
    raw = bases.tobytes()
    body = eol.join(raw[i:i + lenc] for i in range(0, L, lenc)) + eol
    eng = _ffi.Engine(0)
    cid = eng.add_contig_text(np.frombuffer(body, dtype=np.uint8), L, lenc, lenc + len(eol))
    want = np.frombuffer(raw.upper(), dtype=np.uint8)
    assert np.array_equal(eng.read_contig(cid), want)
    # no trailing newline / body shorter than claimed
    cid2 = eng.add_contig_text(np.frombuffer(body[:-len(eol)], dtype=np.uint8), L, lenc, lenc + len(eol))
    assert np.array_equal(eng.read_contig(cid2), want)
    with pytest.raises(_ffi.MsimError):
        eng.add_contig_text(np.frombuffer(body[:1000], dtype=np.uint8), L, lenc, lenc + len(eol))
    eng.close()


def _splice_numpy(A, B, bp_a, bp_b):
    """it_mutator.py:121-146: pairwise([0] + bp + [len]) over both contigs, even intervals from A, odd ones from B."""
    ca, cb = [0] + [int(x) for x in bp_a] + [len(A)], [0] + [int(x) for x in bp_b] + [len(B)]
    parts = [(B[cb[j]:cb[j + 1]] if j % 2 else A[ca[j]:ca[j + 1]]) for j in range(len(ca) - 1)]
    return np.concatenate(parts)


@pytest.mark.parametrize("la,lb,n_bp,seed", [(50_000, 30_011, 17, 1), (1_000_000, 700_000, 200_000, 2), (3, 9, 1, 3),
                                             (4_000_000, 2_500_000, 1, 4), (100_003, 100_003, 49_000, 5), (64, 1_000_000, 31, 6)])
def test_splice_contigs_equals_numpy(la, lb, n_bp, seed):
    """msim_splice_contigs (k_splice): interchromosomal translocation of one contig, for a few breakpoints and for as many
    as fit (segments of two bases), through the plain and the framed fetch."""
    rs = np.random.RandomState(seed)
    A, B = random_bases(la, seed), random_bases(lb, seed + 50)

    def breakpoints(L):                               # sample_with_minimum_distance(1, L, n_bp, 1): sorted, >= 2 apart
        s = np.sort(rs.choice(np.arange(1, L - (n_bp - 1)), size=n_bp, replace=False))
        return (s + np.arange(n_bp)).astype(np.uint64)
    bp_a, bp_b = breakpoints(la), breakpoints(lb)
    eng = _ffi.Engine(0)
    a, b = eng.add_contig(A), eng.add_contig(B)
    for own, other, x, y, X, Y in ((bp_a, bp_b, a, b, A, B), (bp_b, bp_a, b, a, B, A)):
        cid = eng.splice_contigs(x, y, own, other)
        want = _splice_numpy(X, Y, own, other)
        got = eng.fetch_sequence(cid)
        assert got.shape == want.shape and np.array_equal(got, want)
        for bpl in (60, 7):
            assert eng.fetch_sequence_framed(cid, bpl).tobytes() == _wrap(want, bpl)
    full = eng.splice_contigs(a, -1, np.zeros(0, np.uint64), np.zeros(0, np.uint64))      # __write_chrom_full
    assert np.array_equal(eng.fetch_sequence(full), A)
    with pytest.raises(_ffi.MsimError):               # breakpoints must ascend and lie inside their contig
        eng.splice_contigs(a, b, np.array([5, 3], np.uint64), np.array([1, 2], np.uint64))
    with pytest.raises(_ffi.MsimError):
        eng.splice_contigs(a, b, np.array([1], np.uint64), np.array([lb + 1], np.uint64))
    eng.close()


@pytest.mark.parametrize("L,head", [(40_000_003, 37), (9_000_000, 4096), (300_000, 5), (0, 11)])
def test_text_written_into_the_output_file_equals_the_fetched_text(tmp_path, L, head):
    """msim_fetch_sequence_framed_file / msim_render_vcf_device_file (file_io.hip): the span behind what the file already
    holds receives exactly the bytes the buffer calls return; what was there before and what the caller writes behind the
    span meanwhile stay.  40 Mb is several pieces of the channel's ring, 300 kb less than one."""
    eng = _ffi.Engine(0)
    eng.seed(7, 8)
    eng.set_params(_params(titv=1.0))
    bases = random_bases(L, 31)
    cid = _plan_apply(eng, bases, [_sv_range(0, L - 1, L // 50, ALL_SV, LENS)] if L else [])
    want_fa = eng.fetch_sequence_framed(cid, 60).tobytes() if L else b""
    want_vcf = eng.render_vcf_device(cid, "chrF").tobytes()
    prefix = bytes(range(256)) * 17
    for name, want, call in (("o.fa", want_fa, lambda fd, off: eng.fetch_sequence_framed_to_file(cid, 60, fd, off)),
                             ("o.vcf", want_vcf, lambda fd, off: eng.render_vcf_device_to_file(cid, "chrF", fd, off))):
        if not L and name == "o.fa":
            continue
        with open(tmp_path / name, "w+b") as f:
            f.write(prefix[:head])
            f.flush()
            n = call(f.fileno(), head)
            assert n == len(want)
            f.seek(head + n)
            f.write(b"tail")                          # (the caller goes on behind the span while the channel fills it)
            f.flush()
            eng.file_wait()
        got = (tmp_path / name).read_bytes()
        assert got[:head] == prefix[:head] and got[head:head + len(want)] == want and got[head + len(want):] == b"tail"
    eng.close()


def test_text_into_a_file_that_cannot_be_mapped_is_reported_not_attempted():
    import os
    eng = _ffi.Engine(0)
    eng.seed(1, 1)
    eng.set_params(_params())
    cid = _plan_apply(eng, random_bases(100_000, 2), [])
    r, w = os.pipe()
    try:
        with pytest.raises(_ffi.MsimUnsupported):
            eng.fetch_sequence_framed_to_file(cid, 60, w, 0)
    finally:
        os.close(r)
        os.close(w)
    eng.close()


def test_queued_file_writes_of_many_contigs_land_in_order(tmp_path):
    """Six contigs queued back to back on both channels (a third job waits for the channel's older buffer): the files hold
    every contig's text at its offset once file_wait returns; a write that fails (descriptor opened read-only) is reported by
    file_wait, not lost."""
    eng = _ffi.Engine(0)
    eng.seed(3, 4)
    eng.set_params(_params(titv=1.0))
    wants = []
    with open(tmp_path / "all.fa", "w+b") as fa, open(tmp_path / "all.vcf", "w+b") as vc:
        pos_f = pos_v = 0
        for i in range(6):
            L = 9_000_000 + 1000 * i
            cid = _plan_apply(eng, random_bases(L, 40 + i), [_sv_range(0, L - 1, L // 80, ALL_SV, LENS)])
            wants.append((eng.fetch_sequence_framed(cid, 60).tobytes(), eng.render_vcf_device(cid, f"c{i}").tobytes()))
            pos_f += eng.fetch_sequence_framed_to_file(cid, 60, fa.fileno(), pos_f)
            pos_v += eng.render_vcf_device_to_file(cid, f"c{i}", vc.fileno(), pos_v)
            eng.clear()
        eng.file_wait()
    assert (tmp_path / "all.fa").read_bytes() == b"".join(w[0] for w in wants)
    assert (tmp_path / "all.vcf").read_bytes() == b"".join(w[1] for w in wants)
    cid = _plan_apply(eng, random_bases(100_000, 1), [])
    with open(tmp_path / "all.fa", "rb") as ro:
        eng.fetch_sequence_framed_to_file(cid, 60, ro.fileno(), 0)
        with pytest.raises(_ffi.MsimError, match="pwrite"):
            eng.file_wait()
    eng.file_wait()                                   # (reported once)
    eng.close()


