"""Interchromosomal translocations: the reference's second pass (it_mutator.py:20-220) over the Fasta the mutation pass wrote
(or over the input, in ``it`` mode).

Host side (this file): which contigs take part, who is whose partner, how many breakpoints a pair gets and where they
fall -- a few draws from CPython's ``random`` (``shuffle``, ``choice``, ``sample``), made by that very generator, in the
reference's order, so the stream position and every result agree by construction.  Device side (libmsim): the contigs
go up as file text, ``msim_splice_contigs`` (csrc/text_gpu.hip: ``k_splice``) cuts both at their breakpoints and takes
the segments alternately, the framing kernel wraps the result, and the copy lands in the mapped output file."""
from __future__ import annotations

import os
import random

import numpy as np

from . import _ffi
from ._ffi import Engine as _HostEngine            # (the host-only context that samples: no GPU involved)
from .bedpe_writer import BedpeWriter
from .fasta_writer import FastaWriter
from .util import print_warning


NATIVE_FILE_EGRESS = os.environ.get("MSIM_PY_EGRESS") != "1"   # (diagnosis: 1 = map the output spans here, copy with one thread)
_SAMPLER = None                   # libmsim host-only context of sample_with_minimum_distance (made on first use)
RESIDENT_MAX_BYTES = 64 << 30     # inputs kept in HBM for their partner's turn (a human genome is 3 GB; the GPU holds 288)
RESIDENT_MAX_CONTIGS = 30000      # ... and contig slots of the context (libmsim: 65536; every splice result takes one)
NATIVE_SAMPLE_FROM = 512          # breakpoints per contig from which libmsim's sampler takes over (same draws, ~100x faster)


def sample_with_minimum_distance(start: int, stop: int, k: int, d: int) -> np.ndarray:
    """k sorted positions in [start, stop) that keep at least d between neighbours (reference util.py:93-109): a plain
    ``random.sample`` of a range shortened by (k - 1) * d, sorted, the r-th smallest moved up by r * d.  Raises
    CPython's own ValueError when the range is too short for k.

    A handful of breakpoints are drawn by CPython itself.  A pair of human chromosomes at rate 0.001 gets a quarter of a
    million per contig -- a second of ``random.sample`` each: those go through ``msim_sample_min_distance`` (libmsim's
    host sampler on a host-only context: the generator's state goes in, the same words are consumed, the state comes
    back; pinned against CPython in tests/test_it_host.py)."""
    if k < NATIVE_SAMPLE_FROM:
        picked = random.sample(range(start, stop - (k - 1) * d), k)
        out = np.sort(np.asarray(picked, dtype=np.int64))
        return out + d * np.arange(k, dtype=np.int64)
    from .mutator import sample_setsize
    global _SAMPLER
    st = random.getstate()
    if _SAMPLER is None:                               # one host-only context for every call of the run
        _SAMPLER = _HostEngine(device=-1)
    eng = _SAMPLER
    eng.set_mt_state(0, np.array(st[1][:624], dtype=np.uint32), st[1][624])
    out = eng.sample_min_distance(start, stop, k, d, sample_setsize(k))          # (ValueError: nothing was drawn)
    mt, pos = eng.get_mt_state(0)
    random.setstate((st[0], tuple(mt.tolist()) + (int(pos),), st[2]))
    return out


class ITMutator:
    def __init__(self, args, fasta, sim):
        self._args = args
        self._fasta = fasta
        self._sim = sim
        self._fasta_writer = FastaWriter(args.outfastait)
        self._bedpe_writer = BedpeWriter(args.outbedpe)
        self._eng = None
        self._turns_left: set = set()
        self._resident: dict = {}                      # contig number -> libmsim contig id of its input, while in HBM
        self._resident_bytes = 0
        self._slots = 0                                # contig slots of the context in use
        self._assign_partners(self._available())

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def close(self):
        # what the pass queued for the Fasta (libmsim's output channel) is joined BEFORE the writers close -- also when mutate()
        # raised half-way: a write that failed (ENOSPC ...) is reported, after the teardown, like Mutator.close() does
        eng, self._eng = getattr(self, "_eng", None), None
        pending = None
        if eng is not None and getattr(eng, "h", None):
            try:
                eng.file_wait()
            except Exception as e:  # noqa: BLE001
                pending = e
        try:
            for w in (getattr(self, "_fasta_writer", None), getattr(self, "_bedpe_writer", None)):
                if w is not None:
                    w.close()
        finally:
            if eng is not None:
                eng.close()
        if pending is not None:
            raise pending

    # ------------------------------------------------------------------ who with whom (it_mutator.py:50-83)
    def _available(self) -> list:
        """Contigs with an it rate (0 counts, None does not) and more than two bases."""
        return [c.number for c in self._sim.chromosomes if c.it_rate is not None and len(self._fasta[c.number]) > 2]

    def _assign_partners(self, avail: list):
        """A shuffle, then every contig the walk reaches picks a random partner among those left.  The reference walks
        the very list it removes from, so the walk skips: after taking the contig at index i (and a partner) out, index
        i + 1 is looked at next.  Kept as is -- which contigs stay single depends on it."""
        self._partners = {}
        random.shuffle(avail)
        # (the list as "the slots of the shuffled order that are still there": an assembly has 10^5 contigs, and
        #  list.remove() made this quadratic -- 20 s of it.  Same elements at the same indices, same draws.)
        try:
            from sortedcontainers import SortedList
            left = SortedList(range(len(avail)))       # slots still in the list, ascending = list order
        except ImportError:                            # plain lists do the same, slower
            left = None
        if left is None:
            i = 0
            while i < len(avail):
                chrom = avail[i]
                i += 1
                avail.remove(chrom)
                if avail:
                    partner = random.choice(avail)
                    self._partners[partner] = chrom
                    self._partners[chrom] = partner
                    avail.remove(partner)
            return
        i = 0
        while i < len(left):
            slot = left[i]
            i += 1
            left.remove(slot)
            chrom = avail[slot]
            if left:
                pslot = left[random.randrange(len(left))]      # random.choice(seq) is seq[_randbelow(len(seq))]
                left.remove(pslot)
                partner = avail[pslot]
                self._partners[partner] = chrom
                self._partners[chrom] = partner

    def _pairs_once(self) -> list:
        """One contig of every pair, in the order the pairs were made (it_mutator.py:85-94)."""
        gone = set()                                   # (list.remove() per pair: quadratic over an assembly's contigs)
        for chrom, partner in self._partners.items():
            if chrom not in gone:
                gone.add(partner)
        return [c for c in self._partners if c not in gone]

    # ------------------------------------------------------------------ breakpoints (it_mutator.py:96-119, 158-190)
    def _warn(self, text: str):
        if not self._args.ignore_warnings:
            print_warning(text, self._args.no_color)

    def _breakpoints_of_pair(self, chrom: int):
        partner = self._partners[chrom]
        len1, len2 = len(self._fasta[chrom]), len(self._fasta[partner])
        rate1, rate2 = self._sim.chromosomes[chrom].it_rate, self._sim.chromosomes[partner].it_rate
        amount = int((len1 + len2 - 4) / 2 * ((rate1 + rate2) / 2))
        bp1 = bp2 = np.zeros(0, dtype=np.int64)
        try:                                           # one base at least stays between two breakpoints
            bp1 = sample_with_minimum_distance(1, len1, amount, 1)
            bp2 = sample_with_minimum_distance(1, len2, amount, 1)
        except ValueError:
            self._warn(f"Interchromosomal translocation rate too high for sequence {chrom+1} and {partner+1}.")
        return bp1, bp2

    def _generate_all_breakpoints(self) -> dict:
        out = {}
        for chrom in self._pairs_once():
            partner = self._partners[chrom]
            bp1, bp2 = self._breakpoints_of_pair(chrom)
            if len(bp1) and len(bp2):
                out[chrom] = (bp1, bp2)
                out[partner] = (bp2, bp1)
            else:
                self._warn(f"No interchromosomal translocations could be generated between sequence {chrom+1} and "
                           f"{partner+1} (it rates too low).")
        return out

    # ------------------------------------------------------------------ output (it_mutator.py:121-156, 192-220)
    def _engine(self):
        if self._eng is None:
            self._eng = _ffi.Engine(getattr(self._args, "device", 0) or 0)
        return self._eng

    def _ingest(self, eng, rec, number=None) -> int:
        """The contig's input in HBM.  A contig with a partner is read twice -- at its own turn and at its partner's: it is
        uploaded once and stays resident in between (``number``: its key; ``_forget`` drops everything)."""
        if number is not None and number in self._resident:
            return self._resident[number]
        if getattr(rec, "uniform", False):            # file text -> HBM: strip + upper-case on the device
            cid = eng.add_contig_text(rec.body, len(rec), rec.lenc, rec.lenb)
        else:
            cid = eng.add_contig(rec.bases)
        self._slots += 1
        if number is not None:
            self._resident[number] = cid
            self._resident_bytes += len(rec)
        return cid

    def _forget(self, eng):
        eng.clear()
        self._resident = {}
        self._resident_bytes = 0
        self._slots = 0

    def _untouched_runs(self, n: int, breakpoints: dict) -> list:
        """[a, b) spans of >= 2 consecutive contigs that keep their sequence and are small: an assembly's thousands of
        scaffolds below any breakpoint go through libmsim in ONE pass per span (msim_batch_run with no ranges: a contig on
        its own costs ~0.4 ms of launches and round trips).  Needs the loader's index table; [] otherwise."""
        from .mutator import BATCH_MAX_BASES, BATCH_MAX_CONTIGS, BATCH_MAX_LEN
        tab = getattr(self._fasta, "index_table", None)
        if tab is None or len(tab) != n or getattr(self._fasta, "text", None) is None:
            return []
        lens = tab["n_bases"].astype(np.int64)
        ok = ((tab["flags"] & _ffi.FASTA_NONUNIFORM) == 0) & (lens > 0) & (lens <= BATCH_MAX_LEN) & (tab["lenc"] > 0)
        if breakpoints:
            ok[np.fromiter(breakpoints.keys(), dtype=np.int64)] = False
        cum = np.concatenate(([0], np.cumsum(lens)))
        edges = np.flatnonzero(np.diff(np.concatenate(([0], ok.astype(np.int8), [0]))))
        out = []
        for a, b in zip(edges[0::2].tolist(), edges[1::2].tolist()):
            i = a
            while b - i >= 2:
                j = min(int(np.searchsorted(cum, cum[i] + BATCH_MAX_BASES, side="right")) - 1, b, i + BATCH_MAX_CONTIGS)
                if j - i < 2:
                    break
                out.append((i, j))
                i = j
        return out

    def _copy_run(self, eng, a: int, b: int):
        """Contigs [a, b) as they are (upper-cased, re-wrapped at their own line width), each with its defline TWICE -- the
        reference's __write_chrom_full writes the header again (it_mutator.py:148-156 after :199-202)."""
        from .mutator import params_descriptor
        fa = self._fasta
        tab = fa.index_table[a:b]
        n = b - a
        text = fa.text
        heads = [text[int(h0):int(h1)].tobytes() for h0, h1 in zip(tab["h0"], tab["h1"])]
        blob = np.frombuffer(b"".join(h + b"\n>" + h for h in heads), dtype=np.uint8)
        hlen = np.fromiter((2 * len(h) + 2 for h in heads), dtype=np.int64, count=n)
        hoff = np.concatenate(([0], np.cumsum(hlen)[:-1]))
        t = np.zeros(n, dtype=_ffi.BATCH_CONTIG_DTYPE)
        t["body"] = text.ctypes.data + tab["b0"]
        t["body_bytes"] = tab["b1"] - tab["b0"]
        t["n_bases"] = tab["n_bases"]
        t["lenc"] = tab["lenc"]
        t["lenb"] = tab["lenb"]
        t["header"] = blob.ctypes.data + hoff.astype(np.uint64)
        t["header_len"] = hlen
        t["name"] = t["header"]                        # (no VCF lines: the name is never read, it only must not be NULL)
        t["name_len"] = 1
        eng.set_params(params_descriptor(self._sim))
        self._forget(eng)                              # (a batch takes the context's contig table for itself)
        n_text, _, _, last_line = eng.batch_run_table(t, keep=(blob, text), defer_fasta=True)
        done = False
        if NATIVE_FILE_EGRESS:
            try:                                       # framed by libmsim's host threads, written by its output channel
                fd, pos = self._fasta_writer.native_records_span(n_text)
                eng.batch_fetch_to_files(fd if n_text else -1, pos, -1, 0)
                self._fasta_writer.commit_native_records(pos, n_text, int(tab["lenc"][-1]), last_line)
                done = True
            except _ffi.MsimUnsupported:
                pass
        if not done:
            region = self._fasta_writer.map_records(n_text)
            try:
                if n_text:
                    eng.batch_fetch_fasta(region.view)
            finally:
                self._fasta_writer.commit_records(region, int(tab["lenc"][-1]), last_line)
        del blob

    def _mutate_sequence(self, breakpoints: dict):
        eng = self._engine()
        chroms = self._sim.chromosomes
        in_order = all(chroms[k].number == k for k in range(len(chroms)))
        runs = dict(self._untouched_runs(len(chroms), breakpoints)) if in_order else {}
        k = 0
        while k < len(chroms):
            if k in runs:
                self._copy_run(eng, k, runs[k])
                k = runs[k]
            else:
                self._one_contig(eng, chroms[k], breakpoints)
                k += 1

    def _one_contig(self, eng, chrom, breakpoints: dict):
        none = np.zeros(0, dtype=np.uint64)
        rec = self._fasta[chrom.number]
        bpl = self._fasta.faidx.index[rec.name].lenc
        self._fasta_writer.set_bpl(bpl)
        self._fasta_writer.write_header(rec.long_name)
        paired = chrom.number in breakpoints
        a = self._ingest(eng, rec, chrom.number if paired else None)
        if paired:
            p = self._partners[chrom.number]
            partner = self._fasta[p]
            own, other = breakpoints[chrom.number]
            cid = eng.splice_contigs(a, self._ingest(eng, partner, p), own.astype(np.uint64), other.astype(np.uint64))
            self._bedpe_writer.write(rec.name, own, len(rec), partner.name, other, len(partner))
            self._turns_left.discard(chrom.number)
        else:
            # (the reference's __write_chrom_full writes the header again, it_mutator.py:148-156 after :199-202: a contig
            #  without breakpoints carries its defline twice.  Kept: the files are compared byte by byte)
            self._fasta_writer.write_header(rec.long_name)
            cid = eng.splice_contigs(a, -1, none, none)
        if bpl > 0:
            n_done = None
            if NATIVE_FILE_EGRESS:
                fd, pos = self._fasta_writer.native_span()
                try:                                   # queued on libmsim's output channel (csrc/file_io.hip); mutate() joins
                    n_done = eng.fetch_sequence_framed_to_file(cid, bpl, fd, pos)
                except _ffi.MsimUnsupported:
                    n_done = None
            if n_done is not None:
                q, r = divmod(n_done, bpl + 1)
                self._fasta_writer.commit_native(pos, n_done, q * bpl + r)
            else:
                n_text = eng.fetch_sequence_framed_size(cid, bpl)
                region = self._fasta_writer.map_region(n_text)
                try:
                    eng.fetch_sequence_framed_into(cid, bpl, region.view)
                finally:
                    q, r = divmod(n_text, bpl + 1)
                    self._fasta_writer.commit_region(region, q * bpl + r)
        else:
            self._fasta_writer.write_array(eng.fetch_sequence(cid))
        # the result has left for the file (or sits in the channel's own buffer); the inputs stay while a partner's turn
        # is still to come
        self._slots += 1
        left = self._turns_left
        waiting = any(k in left or self._partners.get(k) in left for k in self._resident)
        if not waiting or self._resident_bytes > RESIDENT_MAX_BYTES or self._slots > RESIDENT_MAX_CONTIGS:
            self._forget(eng)
        else:
            eng.release_result(cid)

    def mutate(self):
        breakpoints = self._generate_all_breakpoints()
        self._turns_left = set(breakpoints)            # paired contigs whose own turn is still to come
        self._mutate_sequence(breakpoints)
        if self._eng is not None:
            self._eng.file_wait()                      # what the output channel still holds goes out
