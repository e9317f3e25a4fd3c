"""Interchromosomal translocations (the reference's second pass: it_mutator.py, bedpe_writer.py) -- CPU tier.

* the oracle's restatement (``Oracle.it_pass``) is pinned against what the REAL reference wrote for the ``it_*`` cases
  of tests/golden/cases (Fasta with its doubled deflines, BEDPE, warnings, stream position);
* the product's host side (``ITMutator``: who with whom, breakpoints, BEDPE, writer calls, CLI flow) runs here against the
  same goldens with the device replaced by a numpy stand-in for the five engine calls it makes -- the device side
  (``msim_splice_contigs`` / ``k_splice``, framing, mapped egress) is the GPU tier's (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import random

import numpy as np
import pytest

from helpers import CASES, all_case_names, case_input_bytes, case_meta, parse_fasta_bytes
from oracle import oracle as orc
from pipeline import run_product_case

IT_CASES = [n for n in all_case_names() if n.startswith("it_")]
IT_ONLY = [n for n in IT_CASES if "fasta_len" not in case_meta(n)]


def _warnings_text(ws):
    return "".join(f"WARNING: {w}\n" for w in ws)


@pytest.mark.parametrize("name", IT_ONLY)
def test_oracle_it_pass_matches_reference_golden(name):
    meta = case_meta(name)
    contigs = parse_fasta_bytes(case_input_bytes(meta))
    o = orc.Oracle()
    o.seed(meta["seed_py"], meta["seed_np"])
    fa, bedpe, ws = o.it_pass(contigs, [c["it_rate"] for c in meta["sim"]["chromosomes"]])
    assert fa == (CASES / name / "expected_ms_it.fa").read_bytes()
    assert bedpe == (CASES / name / "expected_ms_it.bedpe").read_bytes()
    assert _warnings_text(ws) == meta["stderr"]
    assert o.py_words32(4) == meta["py_next_words_after"]


def test_oracle_mutation_pass_then_it_pass_matches_reference_golden():
    """RMT with mutations and it rates: the IT pass reads the mutated Fasta, its draws follow the mutation pass's."""
    name = "it_rmt_mutations"
    meta = case_meta(name)
    contigs = parse_fasta_bytes(case_input_bytes(meta))
    o = orc.Oracle()
    o.seed(meta["seed_py"], meta["seed_np"])
    fa, vcf, _, _ = o.run_genome(contigs, meta["sim"], meta["infile_name"])
    assert fa == (CASES / name / "expected_ms.fa").read_bytes()
    mutated = parse_fasta_bytes(fa)
    for c, src in zip(mutated, meta["contigs"]):
        c["lenc"] = src["lenc"]                      # (the mutated file keeps every record's line width)
    fa_it, bedpe, ws = o.it_pass(mutated, [c["it_rate"] for c in meta["sim"]["chromosomes"]])
    assert fa_it == (CASES / name / "expected_ms_it.fa").read_bytes()
    assert bedpe == (CASES / name / "expected_ms_it.bedpe").read_bytes()
    assert _warnings_text(ws) == meta["stderr"]
    assert o.py_words32(4) == meta["py_next_words_after"]


class NumpyEngine:
    """Test double for the engine calls ``ITMutator`` makes (same meaning as the C-ABI entries they bind)."""

    def __init__(self, device=0, flags=0):
        self.contigs = []

    def add_contig_text(self, body, n_bases, lenc, lenb):
        raw = bytes(np.asarray(body, dtype=np.uint8)).replace(b"\r", b"").replace(b"\n", b"").upper()
        assert len(raw) >= n_bases
        self.contigs.append(np.frombuffer(raw[:n_bases], dtype=np.uint8).copy())
        return len(self.contigs) - 1

    def add_contig(self, bases):
        self.contigs.append(np.asarray(bases, dtype=np.uint8).copy())
        return len(self.contigs) - 1

    def splice_contigs(self, a, b, bp_a, bp_b):
        A = self.contigs[a]
        if len(bp_a) == 0:
            self.contigs.append(A.copy())
            return len(self.contigs) - 1
        B = self.contigs[b]
        ca = [0] + [int(x) for x in bp_a] + [len(A)]
        cb = [0] + [int(x) for x in bp_b] + [len(B)]
        parts = [(B[cb[j]:cb[j + 1]] if j % 2 else A[ca[j]:ca[j + 1]]) for j in range(len(ca) - 1)]
        self.contigs.append(np.concatenate(parts))
        return len(self.contigs) - 1

    def fetch_sequence(self, cid):
        return self.contigs[cid]

    def fetch_sequence_framed_size(self, cid, bpl):
        n = len(self.contigs[cid])
        return n + n // bpl

    def fetch_sequence_framed_into(self, cid, bpl, view):
        s = bytes(self.contigs[cid])
        text = b"".join(s[i:i + bpl] + (b"\n" if len(s[i:i + bpl]) == bpl else b"") for i in range(0, len(s), bpl))
        view[:len(text)] = np.frombuffer(text, dtype=np.uint8)

    def set_params(self, params):
        pass

    def batch_run_table(self, table, keep=(), defer_fasta=False):
        """msim_batch_run for contigs without ranges: the complete run of records, each '>' header line + wrapped body."""
        import ctypes as C
        assert defer_fasta and (table["n_ranges"] == 0).all()
        out = bytearray()
        partial = False
        for row in table:
            body = C.string_at(int(row["body"]), int(row["body_bytes"]))
            head = C.string_at(int(row["header"]), int(row["header_len"]))
            raw = body.replace(b"\r", b"").replace(b"\n", b"").upper()[:int(row["n_bases"])]
            bpl = int(row["lenc"])
            if partial:
                out += b"\n"
            out += b">" + head + b"\n"
            out += b"".join(raw[i:i + bpl] + (b"\n" if len(raw[i:i + bpl]) == bpl else b"") for i in range(0, len(raw), bpl))
            partial = len(raw) % bpl != 0
            last = len(raw) % bpl
        self._batch_text = bytes(out)
        return len(out), np.zeros(0, dtype=np.uint8), np.zeros(len(table), dtype=bool), last

    def batch_fetch_fasta(self, dst):
        dst[:len(self._batch_text)] = np.frombuffer(self._batch_text, dtype=np.uint8)

    # the *_to_file calls (libmsim: queued on an output channel; here: written at once)
    def fetch_sequence_framed_to_file(self, cid, bpl, fd, offset):
        import os
        buf = np.empty(self.fetch_sequence_framed_size(cid, bpl), dtype=np.uint8)
        self.fetch_sequence_framed_into(cid, bpl, buf)
        os.pwrite(fd, buf.tobytes(), offset)
        return len(buf)

    def batch_fetch_to_files(self, fasta_fd, fasta_offset, vcf_fd, vcf_offset):
        import os
        assert vcf_fd == -1
        if fasta_fd >= 0:
            os.pwrite(fasta_fd, self._batch_text, fasta_offset)

    def file_wait(self):
        pass

    def release_result(self, cid):
        self.contigs[cid] = None

    def clear(self):
        self.contigs = []

    def close(self):
        pass


@pytest.mark.parametrize("name", IT_ONLY)
def test_it_mode_cli_host_side_matches_reference_golden(name, tmp_path, monkeypatch):
    """``mutation-simulator file it <rate>`` through the product's CLI with the device stood in for: partner assignment,
    breakpoints, the doubled defline of untouched contigs, BEDPE lines, warnings, the generator's position."""
    from mutation_simulator_amd import _ffi
    monkeypatch.setattr(_ffi, "Engine", NumpyEngine)
    monkeypatch.setattr(_ffi, "warm_up_async", lambda device=0, pin=False: None)
    meta = case_meta(name)
    res = run_product_case(meta, tmp_path)
    assert res["exception"] is None and res["exit_code"] is None, (res["exception"], res["stderr"])
    assert res["fasta"] is None and res["vcf"] is None              # no mutation pass: no _ms files
    assert res["it_fasta"] == (CASES / name / "expected_ms_it.fa").read_bytes()
    assert res["bedpe"] == (CASES / name / "expected_ms_it.bedpe").read_bytes()
    assert res["stderr"] == meta["stderr"]
    assert [random.getrandbits(32) for _ in range(4)] == meta["py_next_words_after"]


def test_bedpe_writer_lines():
    """bedpe_writer.py:44-55: a line per pair of breakpoints, a closing line to both contig ends for an odd count; the
    header line exists as a method the reference never calls."""
    import mutation_simulator_amd as msa
    import tempfile
    from pathlib import Path
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / "x.bedpe"
        w = msa.BedpeWriter(p)
        w.write("c1", [5, 9, 30], 100, "c2", [2, 7, 11], 50)
        w.write("c2", np.array([2, 7]), 50, "c1", np.array([5, 9]), 100)
        w.write("c3", [], 10, "c4", [], 10)
        w.close()
        assert p.read_text() == "c1\t5\t9\tc2\t2\t7\nc1\t30\t100\tc2\t11\t50\nc2\t2\t7\tc1\t5\t9\n"
        with pytest.raises(msa.BedpeWriterError):
            msa.BedpeWriter(Path(td) / "no_such_dir" / "x.bedpe")


def test_sample_with_minimum_distance_is_the_references():
    """util.py:93-109 on CPython's own generator, against the oracle's restatement on its own MT19937."""
    from mutation_simulator_amd.it_mutator import sample_with_minimum_distance
    for seed, (start, stop, k, d) in enumerate([(1, 1000, 40, 1), (1, 90, 44, 1), (0, 5000, 7, 30), (1, 50, 0, 1)]):
        random.seed(seed)
        got = sample_with_minimum_distance(start, stop, k, d)
        o = orc.Oracle()
        o.seed(seed, 0)
        want = o.sample_min_dist(start, stop, k, d)
        assert got.tolist() == [int(x) for x in want]
        assert np.all(np.diff(got) > d) if k > 1 else True
    random.seed(3)
    with pytest.raises(ValueError):
        sample_with_minimum_distance(1, 100, 51, 1)


@pytest.mark.parametrize("start,stop,k,d", [(1, 2_000_000, 50_000, 1), (1, 3_000, 1_400, 1), (1, 250_000, 600, 1),
                                            (0, 90_000, 30_000, 2), (1, 5_000_000, 512, 1)])
def test_native_breakpoint_sampler_equals_cpython(start, stop, k, d, monkeypatch):
    """Above NATIVE_SAMPLE_FROM breakpoints the draws go through msim_sample_min_distance (set path with its duplicate
    handling, pool path for dense samples): same positions, same stream position as CPython's own random.sample."""
    from mutation_simulator_amd import it_mutator
    assert k >= it_mutator.NATIVE_SAMPLE_FROM
    random.seed(1234 + k)
    got = it_mutator.sample_with_minimum_distance(start, stop, k, d)
    after = [random.getrandbits(32) for _ in range(4)]
    monkeypatch.setattr(it_mutator, "NATIVE_SAMPLE_FROM", 1 << 62)
    random.seed(1234 + k)
    want = it_mutator.sample_with_minimum_distance(start, stop, k, d)
    assert got.tolist() == want.tolist()
    assert after == [random.getrandbits(32) for _ in range(4)]
    monkeypatch.undo()
    random.seed(5)
    before = random.getstate()
    with pytest.raises(ValueError, match="Sample larger than population or is negative"):
        it_mutator.sample_with_minimum_distance(1, 2_000, 1_001, 1)
    assert random.getstate() == before               # nothing was drawn


@pytest.mark.parametrize("start,stop,k,d", [(1, (1 << 32) + 12_345, 700, 1), (5, 3 * (1 << 33) + 7, 4_000, 3),
                                            ((1 << 40), (1 << 40) + (1 << 35), 1_000, 2)])
def test_native_sampler_over_ranges_of_2_to_the_32_and_more(start, stop, k, d):
    """random.sample over range(n) with n >= 2**32 takes CPython's multi-word getrandbits (low word first, the last
    word shifted).  libmsim's host sampler -- msim_sample_min_distance, the IT pass's breakpoint sampler -- does the same:
    positions and the generator's position afterwards equal CPython's own ``sample_with_minimum_distance``."""
    import numpy as np
    from mutation_simulator_amd import _ffi, mutator as mm

    def reference(start, stop, k, d):                      # util.py:93-109
        sampl = random.sample(range(start, stop - (k - 1) * d), k)
        return [s + d * r for r, s in enumerate(sorted(sampl))]
    random.seed(77 + k)
    want = reference(start, stop, k, d)
    after = [random.getrandbits(32) for _ in range(4)]
    random.seed(77 + k)
    eng = _ffi.Engine(device=-1)
    mm.export_python_streams(eng)
    got = eng.sample_min_distance(start, stop, k, d, mm.sample_setsize(k))
    mm.import_python_streams(eng)
    eng.close()
    assert np.asarray(got).tolist() == want
    assert after == [random.getrandbits(32) for _ in range(4)]


def test_partner_walk_equals_the_references_list_walk():
    """``_assign_partners`` keeps the reference's semantics -- a walk over the very list it removes from
    (it_mutator.py:59-71) -- without its quadratic ``list.remove``: same pairs in the same dict order, same one-per-pair list
    (it_mutator.py:73-83), same generator state, for every small size and a few larger ones."""
    from mutation_simulator_amd import it_mutator

    class Bare(it_mutator.ITMutator):
        def __init__(self):
            pass

        def __del__(self):
            pass

    def plain(avail):
        partners = {}
        random.shuffle(avail)
        remain = avail                                  # (the reference's alias)
        for chrom in avail:
            remain.remove(chrom)
            if remain:
                partner = random.choice(remain)
                partners[partner] = chrom
                partners[chrom] = partner
                remain.remove(partner)
        once = list(partners.keys())
        for c, p in partners.items():
            if c in once:
                once.remove(p)
        return partners, once
    for n in list(range(0, 12)) + [33, 100, 1001]:
        for seed in range(4):
            random.seed(seed * 1000 + n)
            want = plain(list(range(n)))
            state = random.getstate()
            random.seed(seed * 1000 + n)
            b = Bare()
            b._assign_partners(list(range(n)))
            assert list(b._partners.items()) == list(want[0].items())
            assert b._pairs_once() == want[1]
            assert random.getstate() == state
