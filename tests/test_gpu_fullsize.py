"""Parity at BASELINE.json's full sizes, through size-independent properties and through the oracle
on the largest pieces it finishes in seconds."""
from __future__ import annotations

import numpy as np
import pytest

import bench
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from oracle import oracle as orc
from test_gpu_parity import checksum_host, synth_host
from test_host_settings import dump_sim

pytestmark = pytest.mark.gpu

C3 = ["-in", "0.001", "-inmin", "1", "-inmax", "50", "-de", "0.001", "-demin", "1", "-demax", "50",
      "-du", "0.0005", "-dumin", "50", "-dumax", "500", "-iv", "0.0005", "-ivmin", "50", "-ivmax", "500"]

# SNP tables of the reference (mutator.py:77, 449-455) for plain A/C/G/T input
TI = {ord("A"): ord("G"), ord("G"): ord("A"), ord("T"): ord("C"), ord("C"): ord("T")}
TV = {ord("A"): b"TC", ord("G"): b"CT", ord("T"): b"GA", ord("C"): b"AG"}


def _snp_lut():
    lut = np.zeros((3, 256), dtype=np.uint8)
    for b in b"ACGT":
        lut[0, b] = TI[b]
        lut[1, b] = TV[b][0]
        lut[2, b] = TV[b][1]
    return lut


def test_config2_full_genome_apply_properties():
    """3 Gb / 24 contigs / 30 M SNPs: the mutated stream differs from the input at exactly the record
    positions, by exactly the reference's transition / transversion tables; lengths are unchanged;
    the device checksum equals the host checksum of the fetched stream."""
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.workload_settings(lengths)
    lut = _snp_lut()
    eng = _ffi.Engine(0)
    eng.seed(42, 42)
    eng.set_params(mm.params_descriptor(sim))
    total_recs = 0
    for chrom in sim.chromosomes:
        L = lengths[chrom.number]
        cid = eng.add_contig_synthetic(L, 1000 + chrom.number)
        eng.plan_contig(cid, mm.plan_descriptors(chrom))
        eng.apply_contig(cid)
        out_len, n_rec, n_pool = eng.result_sizes(cid)
        assert out_len == L and n_pool == 0
        recs, _ = eng.fetch_records(cid)
        assert np.all(recs["type"] == 1) and np.all(np.diff(recs["pos"].astype(np.int64)) >= 2)
        inp = eng.read_contig(cid)
        out = eng.fetch_sequence(cid)
        diff = np.flatnonzero(out != inp)
        assert np.array_equal(diff, recs["pos"].astype(np.int64))          # every SNP changes its base
        assert np.array_equal(out[diff], lut[recs["aux"], inp[diff]])
        if chrom.number in (0, 21):                                          # largest and smallest
            assert eng.result_checksum(cid) == checksum_host(out)
        total_recs += n_rec
        eng.clear()
    assert total_recs == sum(int(L * 0.01) for L in lengths)
    eng.close()


def _engine_vs_oracle(lengths, extra, tmp_path, seeds=(42, 42), sim=None):
    """PLAN + APPLY + text through libmsim vs the oracle's complete Fasta / VCF bytes."""
    import mutation_simulator_amd as msa
    sim = sim or bench.workload_settings(lengths, extra=extra)
    contigs = [{"name": f"chr{i+1}", "long_name": f"chr{i+1} synthetic", "lenc": 60,
                "bases": synth_host(L, 1000 + i)} for i, L in enumerate(lengths)]
    o = orc.Oracle()
    o.seed(*seeds)
    want_fa, want_vcf, _, _ = o.run_genome(contigs, dump_sim(sim), "synthetic.fa")
    body = b"".join(l + b"\n" for l in want_vcf.split(b"\n") if l and not l.startswith(b"#"))
    eng = _ffi.Engine(0)
    eng.seed(*seeds)
    eng.set_params(mm.params_descriptor(sim))
    fw = msa.FastaWriter(tmp_path / "o.fa")
    got_vcf = []
    for chrom in sim.chromosomes:
        c = contigs[chrom.number]
        cid = eng.add_contig_synthetic(lengths[chrom.number], 1000 + chrom.number)
        eng.plan_contig(cid, mm.plan_descriptors(chrom))
        eng.apply_contig(cid)
        fw.set_bpl(60)
        fw.write_header(c["long_name"])
        fw.write_array(eng.fetch_sequence(cid))
        recs, pool = eng.fetch_records(cid)
        got_vcf.append(_ffi.render_vcf(recs, pool, c["bases"], c["name"]))
        eng.clear()
    fw.close()
    eng.close()
    assert (tmp_path / "o.fa").read_bytes() == want_fa
    assert b"".join(got_vcf) == body


def test_config2_500mb_vs_oracle(tmp_path):
    _engine_vs_oracle([250_000_000, 250_000_000], [], tmp_path)


def test_config3_sv_mix_200mb_vs_oracle(tmp_path):
    _engine_vs_oracle([150_000_000, 50_000_000], C3 + ["-sn", "0.005"], tmp_path)


def test_config4_rmt_240mb_vs_oracle(tmp_path):
    """BASELINE configs[3] shape (gene-blocking RMT: ~3 300 drawing ranges incl. hot / cold ranges and pool-path hot
    spots) at 240 Mb, straight against the ORACLE -- not the host planner: complete Fasta + VCF bytes."""
    lengths = [150_000_000, 90_000_000]
    sim = bench.workload_settings_rmt(lengths, bench.c4_rmt_text(lengths))
    _engine_vs_oracle(lengths, None, tmp_path, sim=sim)


def test_config4_fixture_scaled_vs_oracle(tmp_path):
    """The REAL configs[3] workload -- the interval-merged Homo sapiens blocks of tests/golden/c4_blocks.npz -- cut down to
    its first contigs' first 120 / 90 Mb: complete Fasta + VCF bytes against the oracle."""
    z = np.load(bench.C4_FIXTURE)
    lengths = [120_000_000, 90_000_000]
    what = ("None", "sn 0.05", "sn 0.001", "sn 0.2")
    out = ["std", "it None", "sn 0.01", ""]
    for ci, L in enumerate(lengths):
        out.append(f"chr {ci + 1}")
        for a, e, k in zip(z[f"s{ci}"].tolist(), z[f"e{ci}"].tolist(), z[f"k{ci}"].tolist()):
            if e < L:
                out.append(f"{a}-{e} {what[k]}")
    sim = bench.workload_settings_rmt(lengths, "\n".join(out) + "\n")
    assert sum(len(mm.plan_descriptors(ch)) for ch in sim.chromosomes) > 1500
    _engine_vs_oracle(lengths, None, tmp_path, sim=sim)


# ---------------------------------------------------------------------- BASELINE configs at 3 Gb, straight against the oracle
def _full_genome_vs_oracle(workload, engines, total=3_000_000_000):
    """Every contig of the 3 Gb bench genome through PLAN + APPLY + device text, compared per contig with the ORACLE's
    Fasta body and VCF lines (SHA-256; memory stays at one contig).  The oracle walks both MT19937 streams sequentially
    across all 24 contigs, so one wrong stream cut anywhere shows up in every later contig."""
    import hashlib
    lengths = bench.contig_lengths(total)
    sim = bench.build_settings(workload, lengths)
    dump = dump_sim(sim)
    eng = _ffi.Engine(0)
    eng.seed(42, 42)
    eng.set_params(mm.params_descriptor(sim))
    o = orc.Oracle()
    o.seed(42, 42)
    o.configure(dump)
    by_number = {ch["number"]: ch for ch in dump["chromosomes"]}
    for chrom in sim.chromosomes:
        L = lengths[chrom.number]
        name = f"chr{chrom.number + 1}"
        cid = eng.add_contig_synthetic(L, 1000 + chrom.number)
        eng.plan_contig(cid, mm.plan_descriptors(chrom))
        eng.apply_contig(cid)
        text = eng.fetch_sequence_framed(cid, 60, guess_len=L)
        _, n_rec, _ = eng.result_sizes(cid, applied=False)
        vcf = eng.render_vcf_device(cid, name, guess=n_rec * 48 + 256)
        bases = eng.read_contig(cid)
        eng.clear()
        want_fa, want_vcf, _ = o.mutate_contig_stream(bases, name, f"{name} synthetic", 60, by_number[chrom.number]["ranges"])
        head = len(f">{name} synthetic\n")
        assert hashlib.sha256(memoryview(want_fa)[head:]).digest() == hashlib.sha256(text).digest(), f"Fasta body of {name}"
        assert hashlib.sha256(want_vcf).digest() == hashlib.sha256(vcf).digest(), f"VCF lines of {name}"
        del bases, text, vcf, want_fa, want_vcf
    st = eng.stats()
    assert {k: v for k, v in st.items() if k.startswith("contigs_") and v} == engines
    # both generators end where the oracle's do
    for stream in (0, 1):
        mt, pos = eng.get_mt_state(stream)
        import random
        r = random.Random()
        r.setstate((3, tuple(int(x) for x in mt) + (int(pos),), None))
        omt, opos = o.get_state(stream)
        q = random.Random()
        q.setstate((3, tuple(omt) + (opos,), None))
        assert [r.getrandbits(32) for _ in range(8)] == [q.getrandbits(32) for _ in range(8)]
    eng.close()


def test_config2_full_genome_vs_oracle():
    """BASELINE configs[1] at full size: 3 Gb, 30 M SNPs, titv 2 (SNP sampler engine)."""
    _full_genome_vs_oracle("c2", {"contigs_snp": 24})


def test_config3_full_genome_vs_oracle():
    """BASELINE configs[2] at full size: the full SV mix, all 24 contigs planned AND applied (SV-mix engine)."""
    _full_genome_vs_oracle("c3", {"contigs_svmix": 24})


def test_config4_full_genome_vs_oracle():
    """BASELINE configs[3] at full size with the real fixture: 34 k drawing ranges, pool-path hot spots (host-cut engine)."""
    _full_genome_vs_oracle("c4", {"contigs_hostcut": 24})


def test_config4sv_full_genome_vs_oracle():
    """The configs[3] file with the configs[2] SV mix as its std line (host-chain engine)."""
    _full_genome_vs_oracle("c4sv", {"contigs_hostchain": 24})


def test_readme_flags_720mb_vs_oracle():
    """The reference README's own benchmark flags (every type at 0.01, translocations included) on the bench genome scaled to
    720 Mb: 43 M candidates, TL / TLI on the boundary walk of the SV-mix engine, __link_tls per contig over the device's
    words, TLI records with linked spans -- per contig against the oracle, both streams to the end."""
    _full_genome_vs_oracle("readme", {"contigs_svmix": 24}, total=720_000_000)


def test_config3_full_genome_length_identity():
    """Full SV mix on 3 Gb: L_out = L + sum(ins) + sum(dup) - sum(del) per contig, from the records."""
    lengths = bench.contig_lengths(3_000_000_000)
    sim = bench.workload_settings(lengths, snp=0.005, titv=1.0, extra=C3)
    eng = _ffi.Engine(0)
    eng.seed(42, 42)
    eng.set_params(mm.params_descriptor(sim))
    for chrom in sim.chromosomes:
        if chrom.number not in (0, 7, 23):          # three contigs are enough for the identity; PLAN is chained
            cid = eng.add_contig_synthetic(lengths[chrom.number], 1000 + chrom.number)
            eng.plan_contig(cid, mm.plan_descriptors(chrom))
            eng.clear()
            continue
        L = lengths[chrom.number]
        cid = eng.add_contig_synthetic(L, 1000 + chrom.number)
        eng.plan_contig(cid, mm.plan_descriptors(chrom))
        eng.apply_contig(cid)
        recs, pool = eng.fetch_records(cid)
        ln = recs["stop"].astype(np.int64) - recs["pos"].astype(np.int64) + 1
        t = recs["type"]
        delta = int(ln[t == 2].sum() + ln[t == 4].sum() - ln[t == 3].sum())
        out_len, n_rec, n_pool = eng.result_sizes(cid)
        assert out_len == L + delta and n_pool == int(ln[t == 2].sum())
        # spans never overlap and stay inside the contig
        span_end = np.where((t == 3) | (t == 4) | (t == 5), recs["stop"], recs["pos"]).astype(np.int64)
        assert np.all(recs["pos"][1:].astype(np.int64) > span_end[:-1]) and span_end.max() < L
        eng.clear()
    eng.close()


def test_contig_beyond_the_binned_sampler_vs_oracle():
    """A 1.25 Gb contig, `-sn 0.01 -titv 2.0`: n = 1.2375e9 > 2^30, the reach of the binned sampler (1024 bins of 2^20 values) and of
    the anchored windows -- its sample goes through the global-atomic path on the chain (k_accept_scatter / k_bitmap_insert), its
    12.5 M records through the emission train (76 k tiles, 4.7 k super-blocks of counts) -- between two ordinary contigs, in the
    bench's order, against the ORACLE (Fasta body, VCF text, both stream positions)."""
    from test_gpu_bench_order import _bench_order_vs_oracle
    lengths = [30_000_000, 1_250_000_000, 20_000_003]
    st = _bench_order_vs_oracle("c2", lengths, {"contigs_snp": 3})
    assert st["records"] == sum(int(L * 0.01) for L in lengths)
    # (the first contig starts exact and the second is out of the anchored windows' reach: both sample on the chain; so does the
    #  third -- 200 k draws are fewer than the uncertainty of a start behind 13 M draws and 12.5 M SNPs)
    assert st["snp_samples_ahead"] == 0


def test_contig_of_4_gib_is_refused_not_truncated():
    """a contig of 2^32 bases or more would take CPython's multi-word getrandbits in random.sample (util.py:104) and
    32-bit positions end there: MSIM_ERR_UNSUPPORTED at the door (the host package falls back to nothing -- it says so), never a
    truncated length.  One base less is taken."""
    with _ffi.Engine(0) as eng:
        with pytest.raises(_ffi.MsimUnsupported, match="4 GiB"):
            eng.add_contig_synthetic(1 << 32, 1)
        cid = eng.add_contig_synthetic((1 << 32) - 1, 1)
        assert cid == 0
        sim = bench.workload_settings([(1 << 32) - 1], snp=0.0001)
        eng.seed(3, 3)
        eng.set_params(mm.params_descriptor(sim))
        eng.plan_contig(cid, mm.plan_table(sim.chromosomes[0]))
        eng.apply_contig(cid)
        out_len, n_rec, _ = eng.result_sizes(cid)
        assert out_len == (1 << 32) - 1 and n_rec == int(((1 << 32) - 1) * 0.0001)
        recs, _ = eng.fetch_records(cid)
        assert np.all(np.diff(recs["pos"].astype(np.int64)) >= 2) and int(recs["pos"][-1]) < (1 << 32) - 1


@pytest.mark.parametrize("workload,engine", [("c3", "contigs_svmix"), ("c4", "contigs_hostcut"), ("c4sv", "contigs_hostchain")])
def test_one_1250_mb_contig_through_the_host_chain_engines_vs_oracle(workload, engine):
    """The engines with a host chain on ONE contig five times the largest human chromosome (10 M candidates of the SV mix and a
    mutated length of 1.38 Gb; 14 k RMT ranges on one contig): word windows, accept tables, candidate and offset arrays at sizes
    the 3 Gb genome never reaches per contig -- against the ORACLE, a small contig behind it (the streams must continue exactly)."""
    from test_gpu_bench_order import _bench_order_vs_oracle
    lengths = [1_250_000_000, 10_000_000]
    st = _bench_order_vs_oracle(workload, lengths)
    assert st[engine] >= 1, {k: v for k, v in st.items() if k.startswith("contigs_") and v}
