"""FASTA ingest and the two writers (host side of the drop-in surface)."""
from __future__ import annotations

import numpy as np
import pytest

import mutation_simulator_amd as msa
from helpers import CASES, all_case_names, case_meta, parse_fasta_bytes
from mutation_simulator_amd.fasta_io import Fasta


def test_loader_basic(tmp_path):
    p = tmp_path / "a.fa"
    p.write_bytes(b">c1 first contig\nacgtNN\nACG\n>c2\r\nAAAA\r\nCC\r\n>c3 empty\n>c4\nA")
    f = Fasta(p)
    assert list(f.keys()) == ["c1", "c2", "c3", "c4"]
    assert str(f["c1"]) == "ACGTNNACG" and f[0].long_name == "c1 first contig"
    assert f.faidx.index["c1"].lenc == 6 and f.faidx.index["c2"].lenc == 4
    assert str(f[1]) == "AAAACC" and len(f[2]) == 0 and str(f[3]) == "A"
    assert f[0][2] == "G" and f[0][1:4] == "CGT"
    assert (tmp_path / "a.fa.fai").read_text().splitlines()[0] == "c1\t9\t17\t6\t7"


def test_loader_errors(tmp_path):
    p = tmp_path / "dup.fa"
    p.write_bytes(b">x 1\nAC\n>x 2\nGT\n")
    with pytest.raises(msa.FastaDuplicateHeaderError, match="contains duplicate header"):
        msa.load_fasta(p)
    p = tmp_path / "ragged.fa"
    p.write_bytes(b">x\nACGT\nAC\nACGT\n")
    with pytest.raises(msa.FastaIndexingError):
        msa.load_fasta(p)
    with pytest.raises(msa.FastaNotFoundError):
        Fasta(tmp_path / "missing.fa")


@pytest.mark.parametrize("name", [n for n in all_case_names() if case_meta(n)["store"] == "full"
                                  and (CASES / n / "expected_ms.fa").exists()])
def test_fasta_writer_framing_reproduces_reference_files(name, tmp_path):
    """Feeding the reference's own mutated sequences through our FastaWriter (bulk path, split at
    awkward places) must give the reference's bytes: wrap at the INPUT contig's line width, newline
    before a header only after a partial line, no trailing newline after a partial last line."""
    meta = case_meta(name)
    want = (CASES / name / "expected_ms.fa").read_bytes()
    seqs = parse_fasta_bytes(want)
    out = tmp_path / "o.fa"
    w = msa.FastaWriter(out)
    rs = np.random.RandomState(1)
    for c, g in zip(seqs, meta["contigs"]):
        w.set_bpl(g["lenc"])
        w.write_header(g["long_name"])
        b = c["bases"]
        cuts = sorted(set(int(x) for x in rs.randint(0, len(b) + 1, 5))) if len(b) else []
        prev = 0
        for cut in cuts + [len(b)]:
            w.write_array(b[prev:cut])
            prev = cut
    w.close()
    assert out.read_bytes() == want


def test_fasta_writer_single_base_api(tmp_path):
    out = tmp_path / "o.fa"
    w = msa.FastaWriter(out)
    w.set_bpl(3)
    w.write_header("h1")
    for ch in "ACGTA":
        w.write(ch)
    w.write_header("h2")
    w.write_multi("ACG")
    w.write_header("h3")
    w.write_multi(["T", "T"])
    w.close()
    assert out.read_bytes() == b">h1\nACG\nTA\n>h2\nACG\n>h3\nTT"


def test_fasta_writer_whole_records_continue_like_single_writes(tmp_path):
    """write_records (a run of complete records as libmsim's batch path returns them: header lines included, a newline
    before a header iff the body before it ended mid-line) leaves the writer in the state the per-record calls would:
    whatever follows -- another header, more records -- lands on the same bytes."""
    import numpy as np
    recs = [("a x", b"ACGTACG", 3), ("b", b"", 4), ("c", b"ACGTACGT", 4), ("d", b"AC", 5)]

    def framed(seq, bpl):
        return b"\n".join(seq[i:i + bpl] for i in range(0, len(seq), bpl)) + (b"\n" if seq and len(seq) % bpl == 0 else b"")

    ref = msa.FastaWriter(tmp_path / "ref.fa")
    ref.set_bpl(7)
    ref.write_header("first")
    ref.write_multi("ACGTA")                                  # the run starts behind a partial line
    for name, seq, bpl in recs:
        ref.set_bpl(bpl)
        ref.write_header(name)
        ref.write_framed(np.frombuffer(framed(seq, bpl), dtype=np.uint8), len(seq))
    ref.write_header("after")
    ref.write_multi("GG")
    ref.close()

    text = b""
    for i, (name, seq, bpl) in enumerate(recs):
        if i and recs[i - 1][1] and len(recs[i - 1][1]) % recs[i - 1][2]:
            text += b"\n"
        text += b">" + name.encode() + b"\n" + framed(seq, bpl)
    got = msa.FastaWriter(tmp_path / "got.fa")
    got.set_bpl(7)
    got.write_header("first")
    got.write_multi("ACGTA")
    got.write_records(np.frombuffer(text, dtype=np.uint8), recs[-1][2], len(recs[-1][1]) % recs[-1][2])
    got.write_header("after")
    got.write_multi("GG")
    got.close()
    assert (tmp_path / "got.fa").read_bytes() == (tmp_path / "ref.fa").read_bytes()


def test_vcf_writer_header_and_record(tmp_path):
    name = "snp_titv2_2ctg"
    meta = case_meta(name)
    want = (CASES / name / "expected_ms.vcf").read_bytes()
    inp = tmp_path / meta["infile_name"]
    inp.write_bytes((CASES / name / "input.fa").read_bytes())
    fasta = msa.load_fasta(inp)
    w = msa.VcfWriter(tmp_path / "o.vcf")
    w.write_header(inp.name, fasta, "Unknown", "Unknown", "Unknown")
    w.write(msa.VcfRecord("sn", 5, ref="A", alt="A"), "chrA")          # suppressed: REF == ALT
    w.write(msa.VcfRecord("sn", 7, ref="A", alt="G"), "chrA")
    w.write(msa.VcfRecord("DEL", 9, 12, 3, "ACGT", "A"), "chrA")
    w.close()
    got = (tmp_path / "o.vcf").read_bytes().split(b"\n")
    header_want = [l for l in want.split(b"\n") if l.startswith(b"#")]
    header_got = [l for l in got if l.startswith(b"#")]
    assert [l for l in header_got if not l.startswith(b"##filedate=")] == \
           [l for l in header_want if not l.startswith(b"##filedate=")]
    import datetime
    now = datetime.datetime.now()
    assert f"##filedate={now.year}{now.month}{now.day}".encode() in header_got
    assert got[len(header_got):] == [b"chrA\t7\t.\tA\tG\t.\t.\t.\tGT\t1",
                                     b"chrA\t9\t.\tACGT\tA\t.\t.\tSVTYPE=DEL;END=12;SVLEN=3\tGT\t1", b""]


def test_writer_errors(tmp_path):
    with pytest.raises(msa.FastaWriterError, match="Cannot write to Fasta file"):
        msa.FastaWriter(tmp_path / "nodir" / "x.fa")
    with pytest.raises(msa.VcfWriterError, match="Cannot write to VCF file"):
        msa.VcfWriter(tmp_path / "nodir" / "x.vcf")


# ---------------------------------------------------------------------- the batch path's array formulations
def test_sample_setsize_array_equals_cpythons_float_expression():
    """``sample_setsize_array`` (integers) against ``sample_setsize`` (CPython's ``21 + 4 ** ceil(log(3 k, 4))``) around every
    power of four and on random counts."""
    import numpy as np
    from mutation_simulator_amd import mutator as mm
    ks = list(range(0, 3000))
    for m in range(1, 18):                               # (k < 2**33: contigs are below 4 GiB, k below half of that)
        ks += [4 ** m // 3 + d for d in range(-3, 4)]
    ks += [int(x) for x in np.random.RandomState(3).randint(1, 2 ** 33, size=20000)]
    ks = np.array([k for k in ks if k >= 0], dtype=np.int64)
    assert np.array_equal(mm.sample_setsize_array(ks), np.array([mm.sample_setsize(int(k)) for k in ks]))


def test_units_by_arrays_equal_units_by_contig(tmp_path, monkeypatch):
    """The vectorised cut of the contig loop into batches / single contigs (ARGS mode) against the per-contig greedy rule, on
    assemblies with runs of small contigs, large ones in between, and caps that bite."""
    import numpy as np
    import mutation_simulator_amd as msa
    from mutation_simulator_amd import mutator as mm
    rs = np.random.RandomState(11)
    parts = []
    for i in range(700):
        L = int(rs.choice([0, 1, 50, 900, 5_000, 30_000]))
        if i in (100, 101, 350):
            L = 70_000
        seq = rs.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
        nl = b"\r\n" if i % 97 == 0 else b"\n"
        body = nl.join(seq[a:a + 60] for a in range(0, L, 60))
        if i % 53 == 0 and L > 200:                      # mixed terminators: not batchable
            body = body.replace(b"\n", b"\r\n", 1) if nl == b"\n" else body
        parts.append(b">s%d\n" % i + body + (nl if L else b""))
    fa = tmp_path / "a.fa"
    fa.write_bytes(b"".join(parts))
    monkeypatch.setattr(mm, "BATCH_MAX_LEN", 40_000)
    args = msa.get_args(["-q", "-o", str(tmp_path / "o"), str(fa), "args", "-sn", "0.01"])
    fasta = msa.load_fasta(args.infile)
    sim = msa.SimulationSettings.from_args(args, fasta, True)
    m = msa.Mutator(args, fasta, sim)
    hit_count_cap = hit_bases_cap = False
    for max_contigs, max_bases in ((37, 120_000), (11, 10_000_000)):
        monkeypatch.setattr(mm, "BATCH_MAX_CONTIGS", max_contigs)
        monkeypatch.setattr(mm, "BATCH_MAX_BASES", max_bases)
        lazy = m._units(sim.chromosomes)
        plain = m._units(list(sim.chromosomes))
        # the per-contig greedy rule, spelled out
        chroms, want, i = list(sim.chromosomes), [], 0
        while i < len(chroms):
            j, total = i, 0
            while (j < len(chroms) and j - i < max_contigs and m._batchable(chroms[j])
                   and total + len(fasta[chroms[j].number]) <= max_bases):
                total += len(fasta[chroms[j].number])
                j += 1
            if j - i >= 2:
                want.append((i, j)); i = j
            else:
                want.append((i, i + 1)); i += 1
        assert lazy == plain == want
        hit_count_cap |= any(j - i == max_contigs for i, j in lazy)
        hit_bases_cap |= any(2 <= j - i < max_contigs for i, j in lazy)
    m.close()
    assert hit_count_cap and hit_bases_cap


def test_plan_table_equals_plan_descriptors(tmp_path):
    """``mutator.plan_table`` (array operations over a contig's ranges) against ``plan_descriptors`` (one ``Range`` at a time)
    on an RMT with hot / cold / blocked ranges, SV settings of their own and an unlisted contig."""
    import ctypes as C
    import numpy as np
    import bench
    from mutation_simulator_amd import _ffi, mutator as mm
    lengths = [3_000_000, 900_000, 40_000]
    rs = np.random.RandomState(5)
    rows, at = ["titv = 1.5", "", "std", "it None", bench.C4_STD_SV, "", "chr 1"], 1
    for i in range(400):
        at += int(rs.randint(2, 5000))
        e = at + int(rs.randint(1, 9000))
        if e >= lengths[0]:
            break
        rows.append(f"{at}-{e} " + ("None", "sn 0.05", "sn 0.001 de 0.002 demin 3 demax 60", "sn 0.2 in 0.01 inmin 1 inmax 50")[i % 4])
        at = e + (1 if i % 9 else 0)
    rows += ["chr 3", "100-20000 sn 0.3"]
    sim = bench.workload_settings_rmt(lengths, "\n".join(rows) + "\n")
    for chrom in sim.chromosomes:
        want = mm.plan_descriptors(chrom)
        got = mm.plan_table(chrom)
        assert got.shape[0] == len(want) and got.dtype == _ffi.RANGE_DTYPE
        raw = b"".join(bytes(r) for r in want)
        masked = got.copy()
        masked["_pad"] = 0
        assert masked.tobytes() == raw
    assert sum(len(mm.plan_descriptors(c)) for c in sim.chromosomes) > 300


def test_native_range_tables_equal_python_on_a_million_random_ranges():
    """``msim_build_ranges`` (C: the reference's float expressions with the same IEEE operations) against
    ``mutator.range_descriptor`` (the Python expressions themselves): one million random ranges over random settings -- rates
    from 1e-9 to 0.5 incl. sums that round differently in other orders, chances that do not sum to one, spans from 1 to 2^31 --
    k, setsize and every cdf threshold bit for bit; then the array formulation (``_plan_table_python``) on the same ranges."""
    import numpy as np
    from mutation_simulator_amd import _ffi, mutator as mm
    from mutation_simulator_amd.mut_types import MutType
    rs = np.random.RandomState(11)
    types = list(MutType)

    class MS:
        has_mutations = True

    class RD:
        pass

    class Chrom:
        pass
    sets = []
    for _ in range(8):
        ms = MS()
        picked = [types[i] for i in rs.permutation(len(types))[:int(rs.randint(1, 8))]]
        ms.mut_rates = {t: float(10 ** rs.uniform(-9, -0.7)) * float(rs.rand() < 0.9) for t in picked}
        tot = sum(ms.mut_rates.values()) or 1.0
        ms.mut_chances = {t: r / tot if rs.rand() < 0.7 else float(rs.rand()) for t, r in ms.mut_rates.items()}
        if not any(ms.mut_chances.values()):
            ms.mut_chances[picked[0]] = 1.0
        ms.mut_lengs = {"min": {t: int(rs.randint(1, 60)) for t in picked}, "max": {t: int(rs.randint(60, 10_000)) for t in picked}}
        sets.append(ms)
    n = 1_000_000
    span = np.where(rs.rand(n) < 0.5, rs.randint(1, 5_000, n), (2.0 ** rs.uniform(0, 31, n)).astype(np.int64))
    start = rs.randint(0, 1 << 30, n).astype(np.int64)
    sid = rs.randint(0, len(sets), n)
    chrom = Chrom()
    chrom.range_definitions = []
    for a, w, q in zip(start.tolist(), span.tolist(), sid.tolist()):
        rd = RD()
        rd.start, rd.stop, rd.mutation_settings = a, a + w - 1, sets[q]
        chrom.range_definitions.append(rd)
    got = mm.plan_table(chrom)
    assert got.shape[0] == n
    alt = mm._plan_table_python(chrom)
    g2, a2 = got.copy(), alt.copy()
    g2["_pad"] = 0
    a2["_pad"] = 0
    assert g2.tobytes() == a2.tobytes()
    for i in rs.randint(0, n, 20_000).tolist() + list(range(2000)):        # the Python expressions themselves, range by range
        r = mm.range_descriptor(chrom.range_definitions[i])
        row = got[i]
        assert (row["start"], row["stop"], row["k"], row["setsize"], row["n_types"]) == (r.start, r.stop, r.k, r.setsize, r.n_types)
        assert row["cdf_thr"].tolist() == list(r.cdf_thr) and row["types"].tolist() == list(r.types)
        assert row["min_len"].tolist() == list(r.min_len) and row["max_len"].tolist() == list(r.max_len)

