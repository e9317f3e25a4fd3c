"""Parity ON THE CALL ORDER ``bench.py`` TIMES: plan + apply every contig first, synchronise, read afterwards.

Every other oracle test reads a contig's results right behind its plan (as the CLI does: ``plan_was_empty`` follows
``plan_contig``, mutator.py:513 of this package), which flushes the SNP sampler's emission group of ONE contig and applies every
contig on its own.  ``bench.one_step`` never reads between contigs, so there the emission runs in groups
(``gpu_emit_flush``: ``k_bitmap_count_b`` / ``k_scan_u32_b`` / ``k_snp_emit_abs_b`` / ``k_bitmap_expand_b`` /
``k_tile_index_batch`` / ``k_rewrite_snp_b`` -- the kernel ``roofline.kernel`` names), the host-chain engines defer every APPLY
into the next contig's walk, and the counter-based engine sends a genome as two batches with a head flush.  These tests put the
CPU ORACLE (not the product's own host planner) on exactly those paths:

* the four BASELINE shapes at the full 3 Gb through ``bench.one_step`` itself, then per contig the SHA-256 of the framed Fasta
  body and of the device VCF text against the oracle walking both MT19937 streams across all 24 contigs; c2 must have gone
  through 12 rewrite launches (pairs);
* a 7-contig 300 Mb genome for emission groups of 1, 2, 3 and 4 (the last group partial), against the oracle;
* ``--rng fast``: 24 contigs queued before any read (two batches + head flush, several cycles), exact equality with
  ``tests/fast_twin.py`` per contig and the mutated streams equal to the contig-by-contig order.

Reference: mutator.py:105-142 (the contig loop), :318-426 (``__mutate_sequence``)."""
from __future__ import annotations

import hashlib
import random

import numpy as np
import pytest

import bench
import fast_twin as ft
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm
from oracle import oracle as orc
from test_fast_host import _twin_ranges, assert_plan_equals_twin
from test_gpu_parity import synth_host
from test_host_settings import dump_sim

pytestmark = pytest.mark.gpu


def _sha(b) -> bytes:
    return hashlib.sha256(memoryview(b)).digest()


def _next_words(mt, pos, n=8):
    r = random.Random()
    r.setstate((3, tuple(int(x) for x in mt) + (int(pos),), None))
    return [r.getrandbits(32) for _ in range(n)]


def _bench_order_vs_oracle(workload, lengths, engines=None, launches=None, sim=None, seed=42, owned=None):
    """All contigs resident, ``bench.one_step`` (the function the driver's number is timed on), THEN the reads.
    ``owned``: this "rank" plans + applies only these contigs and walks the others with ``msim_plan_chain`` (the sharded step of
    ``bench.py --gpus N``); the oracle still walks every contig."""
    sim = sim or bench.build_settings(workload, lengths)
    dump = dump_sim(sim)
    everything = list(range(len(lengths)))
    mine = everything if owned is None else list(owned)
    eng = _ffi.Engine(0)
    try:
        cids = [eng.add_contig_synthetic(L, 1000 + i) if i in mine else None for i, L in enumerate(lengths)]
        eng.set_params(mm.params_descriptor(sim))
        shard_lengths = None if owned is None else lengths
        bench.one_step(eng, sim, cids, mine, seed, mm.plan_table, shard_lengths)       # warm-up step, as in the bench
        eng.reset_stats()
        bench.one_step(eng, sim, cids, mine, seed, mm.plan_table, shard_lengths)
        st = eng.stats()
        if engines is not None:
            assert {k: v for k, v in st.items() if k.startswith("contigs_") and v} == engines
        if launches is not None:
            assert st["apply_launches"] == launches, st["apply_launches"]
        o = orc.Oracle()
        o.seed(seed, seed)
        o.configure(dump)
        by_number = {ch["number"]: ch for ch in dump["chromosomes"]}
        for chrom in sim.chromosomes:
            i = chrom.number
            L = lengths[i]
            name = f"chr{i + 1}"
            if i not in mine:                                      # another rank's contig: the oracle's streams move on
                o.mutate_contig_stream(synth_host(L, 1000 + i), name, f"{name} synthetic", 60, by_number[i]["ranges"])
                continue
            text = eng.fetch_sequence_framed(cids[i], 60, guess_len=L)
            _, n_rec, _ = eng.result_sizes(cids[i], applied=False)
            vcf = eng.render_vcf_device(cids[i], name, guess=n_rec * 48 + 256)
            bases = eng.read_contig(cids[i])
            want_fa, want_vcf, _ = o.mutate_contig_stream(bases, name, f"{name} synthetic", 60, by_number[i]["ranges"])
            head = len(f">{name} synthetic\n")
            assert _sha(memoryview(want_fa)[head:]) == _sha(text), f"Fasta body of {name}"
            assert _sha(want_vcf) == _sha(vcf), f"VCF lines of {name}"
            del bases, text, vcf, want_fa, want_vcf
        for stream in (0, 1):                                      # both generators end where the oracle's do
            mt, pos = eng.get_mt_state(stream)
            omt, opos = o.get_state(stream)
            assert _next_words(mt, pos) == _next_words(omt, opos)
        return st
    finally:
        eng.close()


def test_config2_bench_order_full_genome_vs_oracle():
    """BASELINE configs[1] exactly as ``bench.py`` runs it: 24 contigs planned + applied, one synchronisation, then the
    reads.  6 rewrite launches = the emission groups (fours since round 6: anchored windows are every context's default)
    went through ``k_rewrite_snp_b``; 23 of the 24 samples ran off the chain (the first starts at an exact position)."""
    st = _bench_order_vs_oracle("c2", bench.contig_lengths(3_000_000_000), {"contigs_snp": 24}, launches=6)
    assert st["snp_samples_ahead"] == 23


def test_config2_bench_order_round5_schedule_vs_oracle(monkeypatch):
    """The same with round 5's schedule (every sample on the chain, emission pairs, six-launch train): MSIM_NO_AHEAD."""
    monkeypatch.setenv("MSIM_NO_AHEAD", "1")
    st = _bench_order_vs_oracle("c2", bench.contig_lengths(3_000_000_000), {"contigs_snp": 24}, launches=12)
    assert st["snp_samples_ahead"] == 0


@pytest.mark.parametrize("workload,engines", [("c3", {"contigs_svmix": 24}), ("c4", {"contigs_hostcut": 24}),
                                              ("c4sv", {"contigs_hostchain": 24})])
def test_secondary_workloads_bench_order_full_genome_vs_oracle(workload, engines):
    """The secondary lines of the bench (configs[2], configs[3] and the RMT + SV shape) in the bench's order: every APPLY of
    these engines is deferred into a later contig's host walk and goes out in GROUPS OF THREE (one tile-index launch, one
    ``k_rewrite_b<140>`` / ``k_rewrite_snp_b`` launch for three contigs), the last group flushed by the synchronisation."""
    _bench_order_vs_oracle(workload, bench.contig_lengths(3_000_000_000), engines, launches=8)


@pytest.mark.parametrize("group", [1, 2, 3, 4])
def test_emission_groups_vs_oracle(group, monkeypatch):
    """Seven contigs, 300 Mb, -sn 0.01 -titv 2.0: groups of 1 / 2 / 3 / 4 contigs per emission + rewrite launch, the last group
    partial (7 = 3 x 2 + 1 = 2 x 3 + 1 = 4 + 3) and flushed by the synchronisation -- against the ORACLE."""
    monkeypatch.setenv("MSIM_EMIT_GROUP", str(group))              # (read when the context is created)
    lengths = [61_000_000, 23_000_000, 55_000_007, 9_999_999, 70_000_000, 31_000_001, 49_999_993]
    assert sum(lengths) == 300_000_000
    st = _bench_order_vs_oracle("c2", lengths, {"contigs_snp": 7}, launches=-(-7 // group))
    assert st["records"] == sum(int(L * 0.01) for L in lengths)


@pytest.mark.parametrize("workload", ["c3", "c4sv"])
def test_fast_rng_bench_order_equals_twin(workload):
    """``bench.fast_rng_steps``' step: 24 contigs queued (plan + apply) before anything is read -- the cycle goes out as two
    batches, from the second cycle on with the early / head flush judged by the previous cycle -- three cycles in a row.  Every
    contig's records and insert pool equal the numpy twin's for (key, contig ordinal); the mutated streams equal those of a
    contig-by-contig run of the same engine (plan, apply, read, one contig per batch)."""
    rs = np.random.RandomState(5)
    lengths = [int(x) for x in rs.randint(3_000_000, 7_000_000, size=24)]
    sim = bench.build_settings(workload, lengths)
    params = mm.params_descriptor(sim)
    blocks = {t: int(params.block[t]) for t in range(1, 8)}
    tables = [mm.plan_descriptors(ch) for ch in sim.chromosomes]
    key = 42
    # contig by contig
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    eng.set_fast_key(key)
    want = []
    for ch, t in zip(sim.chromosomes, tables):
        cid = eng.add_contig_synthetic(lengths[ch.number], 1000 + ch.number)
        eng.plan_contig(cid, t)
        eng.apply_contig(cid)
        recs, pool = eng.fetch_records(cid)
        twin = ft.plan(lengths[ch.number], _twin_ranges(t), blocks, int(params.ti_lim), key, ch.number)
        assert_plan_equals_twin(recs, pool, eng.plan_was_empty(cid), twin)
        want.append((_sha(recs.tobytes()), _sha(pool.tobytes()), eng.result_checksum(cid), eng.result_sizes(cid)))
        eng.clear()
    eng.close()
    # the bench's order
    eng = _ffi.Engine(0, _ffi.RNG_FAST)
    eng.set_params(params)
    cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
    for cycle in range(3):
        eng.set_fast_key(key)
        for ch in sim.chromosomes:
            eng.plan_contig(cids[ch.number], mm.plan_table(ch))
            eng.apply_contig(cids[ch.number])
        eng.sync()
        got = []
        for ch in sim.chromosomes:
            cid = cids[ch.number]
            recs, pool = eng.fetch_records(cid)
            got.append((_sha(recs.tobytes()), _sha(pool.tobytes()), eng.result_checksum(cid), eng.result_sizes(cid)))
        assert got == want, f"cycle {cycle}"
    st = eng.stats()
    assert st["contigs_fast"] == 72 and st["plan_host_ms"] == 0.0
    eng.close()


def test_bench_cpu_baseline_carries_matches_gpu():
    """``bench.cpu_baseline``'s sample genome goes through ``bench.one_step`` on the GPU and the line says whether the two agree
    (here on a reduced sample; the driver's run uses 4 x 250 Mb)."""
    out = bench.cpu_baseline(40_000_000, "c2", n_contigs=4, device=0)
    assert out["matches_gpu"] is True, out
    assert out["matches_gpu_detail"]["apply_launches"] == 1          # (one emission group of four: k_rewrite_snp_b)
    out = bench.cpu_baseline(24_000_000, "c3", n_contigs=3, device=0)
    assert out["matches_gpu"] is True, out


@pytest.mark.parametrize("train", ["3", "6"])
@pytest.mark.parametrize("sn_block", [1, 300])
def test_emission_train_tile_index_with_ranges_inside_the_contig_vs_oracle(train, sn_block, monkeypatch):
    """The three-launch emission train (outcomes + popcounts, expansion with its own rank base and the APPLY tile index, rewrite)
    where the tile index has edges to get right: RMT contigs whose ONE drawing range starts and ends inside the contig (tile
    borders in front of the first and behind the last bitmap word), a range of one base short of the contig, contigs shorter
    than a tile and of exactly whole tiles, and a sampling distance of 300 (every `*_block` = 300: a bitmap word's positions span several
    16 KiB tiles) -- all in the bench's order (groups of two, applied with their group) against the ORACLE; the six-launch
    train of rounds 4-5 (MSIM_EMIT_TRAIN=6) through the same cases."""
    monkeypatch.setenv("MSIM_EMIT_TRAIN", train)
    lengths = [5_000_000, 3_000_000, 4_000_000, 16_384 * 40, 2_000_003, 1_000_000]
    head = "".join(f"{t}_block = {sn_block}\n" for t in ("sn", "in", "de", "iv", "du", "tl")) if sn_block != 1 else ""
    rate = "0.0016" if sn_block != 1 else "0.01"             # (k >= 4096 on the Mb-sized contigs: the SNP sampler)
    text = (
            head + "std\nit None\nNone\n\n"
            f"chr 1\n1000001-4000000 sn {rate}\n"
            f"chr 2\n20001-2999000 sn {rate}\n"
            f"chr 3\n1-3999999 sn {rate}\n"
            f"chr 4\n1-{16_384 * 40} sn {rate}\n"
            f"chr 5\n500001-2000003 sn {rate}\n"
            f"chr 6\n1-900000 sn {rate}\n")
    sim = bench.workload_settings_rmt(lengths, text)
    st = _bench_order_vs_oracle("c2", lengths, sim=sim)
    assert st["contigs_snp"] >= 3, {k: v for k, v in st.items() if k.startswith("contigs_")}
