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
