"""``Mutator`` -- drop-in for the reference class of the same name (mutator.py:73-479), with the
per-contig work done on an MI355X through libmsim (include/msim.h).

Same constructor, ``mutate()`` and ``close()``; same output files, warnings and exceptions.  What
changed is *where* the work happens:

  reference                                   here
  ---------                                   ----
  random / numpy.random global generators  -> their MT19937 states are handed to libmsim before the
                                              pass and put back, advanced, afterwards, so a caller
                                              that seeded them gets the reference's exact output and
                                              can keep drawing from them as if the reference had run
  __get_mutations per range (Python dicts) -> msim_plan_contig: 16-byte records in HBM
  __mutate_sequence per base               -> msim_apply_contig: HIP rewrite kernel, uint8 stream
  pyfaidx per-base reads                   -> msim_add_contig_text: file text to HBM, stripped + upper-cased there
  FastaWriter.write per base               -> msim_fetch_sequence_framed: line-wrapped on the device, one write
  VcfWriter.write per record               -> msim_render_vcf_device: record lines rendered on the device

All floating-point expressions of the path are evaluated here with the reference's own formulas
(``plan_descriptors``); the C-ABI takes integers only.
"""
from __future__ import annotations

import os
import random
import sys
import time
from math import ceil, log
from typing import Optional

import numpy as np

from . import _ffi
from .fasta_writer import FastaWriter
from .mut_types import MutType
from .util import format_warning
from .vcf_writer import VcfWriter

_TWO53 = 1 << 53


class Mutation:
    """One generated mutation; 0-based inclusive positions (reference mutator.py:26-50)."""
    __slots__ = ("type", "start", "stop", "trans_reverse", "trans_insert_pos")

    def __init__(self, type: MutType, start: int, stop: int = 0, trans_reverse: bool = False,
                 trans_insert_pos: int = 0):
        self.type = type
        self.start = start
        self.stop = stop
        self.trans_reverse = trans_reverse
        self.trans_insert_pos = trans_insert_pos

    def __repr__(self) -> str:
        return (f"{self.type}, {self.start}, {self.stop}, {self.trans_reverse}, "
                f"{self.trans_insert_pos}\n")


# ---------------------------------------------------------------------- settings -> integers
def _ceil_scaled(x: float) -> int:
    """ceil(x * 2**53), exactly."""
    num, den = float(x).as_integer_ratio()
    return -((-num * _TWO53) // den)


def _floor_scaled(x: float) -> int:
    num, den = float(x).as_integer_ratio()
    return (num * _TWO53) // den


def sample_setsize_array(k: np.ndarray) -> np.ndarray:
    """``sample_setsize`` for an int64 array.  In integers: 3k is never a power of 4, so ceil(log(3k, 4)) is the smallest m
    with 4**m > 3k -- half the bit length of 3k - 1, rounded up -- and the float expression of CPython (whose distance
    from the next integer is at least 1 / (3k ln 4), i.e. > 1e-10 for the k < 2**31 a contig below 4 GiB can have) cannot
    land on the other side.  (Beyond k ~ 1e14 it does: the float expression is what counts, so this is for k < 2**33.)"""
    k = np.asarray(k, dtype=np.int64)
    x = np.maximum(3 * k - 1, 1)
    bits = np.frexp(x.astype(np.float64))[1].astype(np.int64)      # bit_length, exact below 2**53
    m = (bits + 1) // 2
    return np.where(k > 5, 21 + (np.int64(1) << (2 * m)), 21).astype(np.int64)


def sample_setsize(k: int) -> int:
    """CPython ``random.sample`` pool/set switch (Lib/random.py), same float expression."""
    setsize = 21
    if k > 5:
        setsize += 4 ** ceil(log(k * 3, 4))
    return setsize


_SETTINGS_CACHE: dict = {}       # settings key -> (settings, rate sum, template bytes): the per-settings part of msim_range


def _settings_key(ms):
    """By VALUE: a settings object changed in place after its first descriptor (tests, callers adjusting rates) must not
    meet its stale template."""
    lengs = ms.mut_lengs or {}
    # (dict items as they are -- MutType members hash and compare by identity: 1 us instead of 6 for four generator
    #  expressions over `.value`, per contig and step in front of every plan)
    return (tuple((ms.mut_rates or {}).items()), tuple((ms.mut_chances or {}).items()),
            tuple((lengs.get("min") or {}).items()), tuple((lengs.get("max") or {}).items()))


def range_descriptor(rd) -> "_ffi.Range":
    """One ``RangeDefinition`` with mutations -> ``msim_range`` (what mutator.py:157-174 derives).  Everything that
    depends only on the range's ``MutationSettings`` (type order, cdf thresholds, length bounds) is computed once per
    distinct settings VALUE -- an assembly with 20 000 scaffolds shares one in ARGS mode -- and copied."""
    ms = rd.mutation_settings
    key = getattr(ms, "_msim_key", None)
    if key is None or key != _settings_key(ms):
        key = _settings_key(ms)
        try:
            ms._msim_key = key                          # (re-derived whenever the values differ from what it was made of)
        except AttributeError:
            pass
    hit = _SETTINGS_CACHE.get(key)
    if hit is None:
        rate_sum = sum(ms.mut_rates.values())                     # mutator.py:160
        t = _ffi.Range()
        chances = list(ms.mut_chances.values())                   # mutator.py:172-173
        # numpy.random.choice(p=...): cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
        cdf = np.cumsum(np.array(chances, dtype=np.float64))
        cdf /= cdf[-1]
        t.n_types = len(chances)
        for j, ty in enumerate(ms.mut_chances):
            t.types[j] = ty.value
            t.cdf_thr[j] = _ceil_scaled(float(cdf[j]))
        if ms.mut_lengs:
            for ty, v in ms.mut_lengs["min"].items():
                t.min_len[ty.value] = v
            for ty, v in ms.mut_lengs["max"].items():
                t.max_len[ty.value] = v
        if len(_SETTINGS_CACHE) > 4096:
            _SETTINGS_CACHE.clear()
        hit = (ms, rate_sum, bytes(t))
        _SETTINGS_CACHE[key] = hit
    r = _ffi.Range.from_buffer_copy(hit[2])
    r.start, r.stop = rd.start, rd.stop
    r.k = int(((rd.stop - rd.start) + 1) * hit[1])                # mutator.py:225
    r.setsize = sample_setsize(r.k)
    return r


def params_descriptor(sim) -> "_ffi.Params":
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    for t, v in sim.mut_block.items():
        p.block[t.value] = v
    titv = sim.titv
    p_ti = titv * (1 / (titv + 1))                                # mutator.py:436
    p.ti_lim = 0 if p_ti != p_ti else min(_floor_scaled(p_ti) + 1, _TWO53)   # `p <= p_ti`
    return p


def plan_descriptors(chrom) -> list:
    return [range_descriptor(rd) for rd in chrom.range_definitions
            if rd.mutation_settings.has_mutations]


def _range_triples(chrom):
    """(start, stop, settings index) arrays of the contig's ranges with mutations + the distinct settings objects: the contig's
    range list in the form ``msim_build_ranges`` takes.  Walking 35 000 ``RangeDefinition`` objects is 20 ms of interpreter
    time per genome, so the arrays are made once per settings tree and kept on the chromosome object (a list that is replaced
    or changes its length is walked again; ranges edited IN PLACE after their first use are not noticed -- nothing in the
    product does that).  The cache holds the list itself, so its identity cannot be reused by a later list; the grouping of the
    ranges BY SETTINGS VALUE is fixed at the first use, the values themselves (rates, lengths) are read on every call
    (``_settings_descs``) -- settings objects edited in place so that two of them stop being equal are not noticed either."""
    rds = chrom.range_definitions
    cached = getattr(chrom, "_msim_triples", None)
    if cached is not None and cached[0][0] is rds and cached[0][1] == len(rds):
        return cached[1:]
    stamp = (rds, len(rds))
    keep = [rd for rd in rds if rd.mutation_settings.has_mutations]
    n = len(keep)
    start = np.fromiter((rd.start for rd in keep), dtype=np.int64, count=n)
    stop = np.fromiter((rd.stop for rd in keep), dtype=np.int64, count=n)
    ids = np.fromiter((id(rd.mutation_settings) for rd in keep), dtype=np.int64, count=n)
    uniq, first, inv = np.unique(ids, return_index=True, return_inverse=True) if n else (ids, ids, ids)
    # an RMT parser makes one settings object per range line: 35 000 objects, three distinct VALUES -- ranges are grouped by value
    by_value, of_object, settings = {}, np.empty(len(uniq), dtype=np.int32), []
    for q, f in enumerate(first.tolist()):
        ms = keep[f].mutation_settings
        key = _settings_key(ms)
        if key not in by_value:
            by_value[key] = len(settings)
            settings.append(ms)
        of_object[q] = by_value[key]
    sid = of_object[inv] if n else np.zeros(0, dtype=np.int32)
    # (the widest range, as Python's int: decides whether any k can leave the integer formulas' domain -- plan_table)
    # (a negative-length filler of overlapping RMT ranges, rmt.py:243-255, makes k negative: then nothing is skipped)
    widest = int((stop - start).max()) + 1 if n and int((stop - start).min()) + 1 >= 0 else (0 if not n else 1 << 62)
    out = (np.ascontiguousarray(start), np.ascontiguousarray(stop), np.ascontiguousarray(sid, dtype=np.int32), settings, widest)
    try:
        chrom._msim_triples = (stamp,) + out
    except AttributeError:
        pass
    return out


def _settings_descs(settings):
    descs = (_ffi.SettingsDesc * max(len(settings), 1))()
    for d, ms in zip(descs, settings):
        d.rate_sum = sum(ms.mut_rates.values())                          # mutator.py:160 (Python's own left-to-right sum)
        chances = ms.mut_chances                                         # mutator.py:172-173
        d.n_types = len(chances)
        for j, (ty, p) in enumerate(chances.items()):
            d.types[j] = ty._value_                                      # (`.value` is a descriptor call; this runs per contig per step)
            d.chances[j] = p
        if ms.mut_lengs:
            for ty, v in ms.mut_lengs["min"].items():
                d.min_len[ty._value_] = v
            for ty, v in ms.mut_lengs["max"].items():
                d.max_len[ty._value_] = v
    return descs


def plan_table(chrom) -> np.ndarray:
    """``plan_descriptors`` as ONE ``msim_range`` table (numpy, ``_ffi.RANGE_DTYPE``), built natively: the contig's ranges as
    (start, stop, settings index) arrays and its few distinct settings go to ``msim_build_ranges`` (csrc/plan_host.cpp), which
    evaluates the reference's float expressions (``int(((stop - start) + 1) * sum(rates))``, ``cumsum(p) / cumsum(p)[-1]``) with
    the same IEEE operations.  Checked against ``range_descriptor`` range by range in the tests (a million random ranges)."""
    rds = chrom.range_definitions
    if len(rds) <= 2:                                   # ARGS mode: one range per contig -- nothing to amortise, no arrays to build
        few = [range_descriptor(rd) for rd in rds if rd.mutation_settings.has_mutations]
        return np.frombuffer(b"".join(bytes(r) for r in few), dtype=_ffi.RANGE_DTYPE) if few else np.zeros(0, dtype=_ffi.RANGE_DTYPE)
    start, stop, sid, settings, widest = _range_triples(chrom)
    n = len(start)
    if n == 0:
        return np.zeros(0, dtype=_ffi.RANGE_DTYPE)
    out = np.empty(n, dtype=_ffi.RANGE_DTYPE)           # (msim_build_ranges writes every byte of every entry, padding included)
    if len(settings) > 8 or any(len(ms.mut_chances) > 8 for ms in settings):
        return _plan_table_python(chrom)
    descs = _settings_descs(settings)
    rc = _ffi.load().msim_build_ranges(descs, len(settings), start.ctypes.data, stop.ctypes.data, sid.ctypes.data, n, out.ctypes.data)
    if rc != _ffi.OK:
        raise _ffi.MsimError(f"msim_build_ranges failed ({rc})")
    # k = int(span * rate_sum) <= widest * max(rate_sum): nothing can leave the domain of the integer formulas (span < 2^53,
    # 0 <= k < 2^33) when the widest range times the largest rate stays below 2^33 and no rate sum is negative or NaN
    if widest < (1 << 53) and all(0.0 <= d.rate_sum and widest * d.rate_sum < float(1 << 33) for d in descs[:len(settings)]):
        return out
    k = out["k"]
    big = (((stop - start) + 1) >= (1 << 53)) | (k >= (1 << 33)) | (k < 0)
    if big.any():                                                        # (outside the integer setsize formula's domain)
        rds = [rd for rd in chrom.range_definitions if rd.mutation_settings.has_mutations]
        for i in np.flatnonzero(big).tolist():
            r = range_descriptor(rds[i])
            out["k"][i], out["setsize"][i] = r.k, r.setsize
    return out


def _plan_table_python(chrom) -> np.ndarray:
    """The same table by numpy array operations (kept as the cross-check of ``msim_build_ranges`` and for contigs with more
    than 8 distinct settings objects)."""
    rds = [rd for rd in chrom.range_definitions if rd.mutation_settings.has_mutations]
    n = len(rds)
    out = np.zeros(n, dtype=_ffi.RANGE_DTYPE)
    if n == 0:
        return out
    start = np.fromiter((rd.start for rd in rds), dtype=np.int64, count=n)
    stop = np.fromiter((rd.stop for rd in rds), dtype=np.int64, count=n)
    ids = np.fromiter((id(rd.mutation_settings) for rd in rds), dtype=np.int64, count=n)
    rate = np.empty(n, dtype=np.float64)
    uniq, first = np.unique(ids, return_index=True)
    for u, f in zip(uniq.tolist(), first.tolist()):
        ms = rds[f].mutation_settings
        range_descriptor(rds[f])                                      # (fills the settings cache)
        _, rate_sum, tmpl = _SETTINGS_CACHE[_settings_key(ms)]
        sel = ids == u
        out[sel] = np.frombuffer(tmpl, dtype=_ffi.RANGE_DTYPE)[0]
        rate[sel] = rate_sum
    out["start"] = start
    out["stop"] = stop
    k = (((stop - start) + 1).astype(np.float64) * rate).astype(np.int64)    # int(((stop - start) + 1) * sum(rates))  mutator.py:225
    big = (((stop - start) + 1) >= (1 << 53)) | (k >= (1 << 33)) | (k < 0)
    out["k"] = k
    out["setsize"] = sample_setsize_array(np.maximum(k, 0))
    for i in np.flatnonzero(big).tolist():                               # (outside the array formulation's domain)
        r = range_descriptor(rds[i])
        out["k"][i], out["setsize"][i] = r.k, r.setsize
    return out


# ---------------------------------------------------------------------- RNG hand-over
def export_python_streams(engine: "_ffi.Engine") -> None:
    """Give libmsim the current states of ``random`` and ``numpy.random``."""
    st = random.getstate()
    engine.set_mt_state(0, np.array(st[1][:624], dtype=np.uint32), st[1][624])
    nst = np.random.get_state()
    engine.set_mt_state(1, np.asarray(nst[1], dtype=np.uint32), int(nst[2]))


def import_python_streams(engine: "_ffi.Engine") -> None:
    """Put the advanced states back so later draws continue where the reference's would."""
    mt, pos = engine.get_mt_state(0)
    st = random.getstate()
    random.setstate((st[0], tuple(int(x) for x in mt) + (int(pos),), st[2]))
    mt, pos = engine.get_mt_state(1)
    nst = np.random.get_state()
    np.random.set_state((nst[0], mt, int(pos), nst[3], nst[4]))


REPLANNED_CONTIGS = 0            # contigs that went through the host planner after a device window overflow (diagnostic)
BATCH_MAX_LEN = int(os.environ.get("MSIM_BATCH_MAX_LEN", 200_000))   # contigs up to this length are batched (below every device PLAN engine's threshold)
BATCH_SPARSE_MAX_LEN = 4_000_000                 # ... longer ones too while they hold few candidates:
BATCH_SPARSE_MAX_CANDIDATES = 8_000              #     a contig on its own costs ~0.6 ms of launches and round trips whatever its size, the batch's
                                                 #     host planner ~40 ns per candidate (measured, -sn 0.01: 300 kb contigs 480 -> 1150 Mbases/s
                                                 #     in batches, 1 Mb contigs the same either way, 3 Mb contigs 1300 vs 960)
BATCH_MAX_BASES = 256 << 20      # bases per batch
NATIVE_FILE_EGRESS = os.environ.get("MSIM_PY_EGRESS") != "1"   # (diagnosis: 1 = map the output spans here, copy with one thread)
BATCH_MAX_CONTIGS = 16384


class Mutator:
    """Runs the mutation pass of one genome on the GPU and writes ``*_ms.fa`` / ``*_ms.vcf``."""

    def __init__(self, args, fasta, sim, engine: Optional["_ffi.Engine"] = None):
        self._args = args
        self._fasta = fasta
        self._sim = sim
        self._fasta_writer = FastaWriter(args.outfasta)
        self._vcf_writer = VcfWriter(args.outvcf)
        self._vcf_writer.write_header(args.infile.name, fasta, sim.assembly_name, sim.species_name,
                                      sim.sample_name)
        self._engine = engine
        self._own_engine = engine is None
        self._t = {"ingest_s": 0.0, "plan_apply_s": 0.0, "fasta_egress_s": 0.0, "vcf_egress_s": 0.0}
        self.stats: dict = {}

    def close(self):
        # the context's teardown (device buffers, registered staging memory) beside the writers' (unmapping the output files)
        pending = None
        if getattr(self._engine, "h", None):           # (also on the reference's KeyError / ValueError: what was queued for the
            try:                                       #  files before it is part of what the reference had written by then)
                self._engine.file_wait()
            except _ffi.MsimError as e:
                pending = e
        eng, self._engine = (self._engine if self._own_engine else None), None
        side = None
        if eng is not None and os.environ.get("MSIM_SERIAL_CLOSE") != "1":
            import threading
            side = threading.Thread(target=eng.close, name="msim-destroy")
            side.start()
        try:
            self._fasta_writer.close()
            self._vcf_writer.close()
        finally:
            if side is not None:
                side.join()
            elif eng is not None:
                eng.close()
        if pending is not None:
            raise pending

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def _fast_rng(self) -> bool:
        """``--rng fast``: libmsim's counter-based generator instead of the reference's streams (msim.h: MSIM_RNG_FAST)."""
        return getattr(self._args, "rng", "compat") == "fast"

    def _open_engine(self, device: int):
        eng = _ffi.Engine(device, _ffi.RNG_FAST if self._fast_rng else _ffi.PLAN_AUTO)
        self._seed_engine(eng)
        return eng

    def _seed_engine(self, eng):
        if self._fast_rng:
            eng.set_fast_key(random.getrandbits(64))       # (from Python's generator: --seed / random.seed() decide the run)
        else:
            export_python_streams(eng)

    def _batchable(self, chrom) -> bool:
        rec = self._fasta[chrom.number]
        if not (getattr(rec, "uniform", False) and 0 < len(rec) <= BATCH_SPARSE_MAX_LEN
                and self._fasta.faidx.index[rec.name].lenc > 0):
            return False
        if len(rec) <= BATCH_MAX_LEN:
            return True
        table = plan_table(chrom)                      # a longer contig: batched while it holds few candidates
        return (int(table["k"].sum()) if len(table) else 0) <= BATCH_SPARSE_MAX_CANDIDATES

    def _warn_empty(self, chrom):
        if not self._args.ignore_warnings:
            print(format_warning(
                f"No mutations could be generated on sequence {chrom.number+1} "
                "(mutation rates too low)", self._args.no_color), file=sys.stderr)

    def _mutate_batch(self, eng, chroms, i, j):
        """Contigs [i, j) of ``chroms`` in ONE pass through libmsim (msim_batch_run)."""
        table = self._batch_table(chroms, i, j)
        if table is not None:
            # the FASTA text is framed straight into the output file's next span (mapped; the deflines sit in the input
            # file's text, which outlives the call)
            fasta, vcf, empty, last_line = eng.batch_run_table(table[0], keep=table[1:], defer_fasta=True)
            bpl_last = int(self._fasta.index_table["lenc"][j - 1])
        else:
            items = []
            for chrom in chroms[i:j]:
                rec = self._fasta[chrom.number]
                items.append((rec.body, len(rec), rec.lenc, rec.lenb, plan_descriptors(chrom), rec.name, rec.long_name))
            fasta, vcf, empty, last_line = eng.batch_run(items)
            last = self._fasta[chroms[j - 1].number]
            bpl_last = self._fasta.faidx.index[last.name].lenc
        for k in np.flatnonzero(empty):
            self._warn_empty(chroms[i + int(k)])
        if isinstance(fasta, int) and NATIVE_FILE_EGRESS:
            # both texts go out on libmsim's output channels while the next batch is planned (csrc/file_io.hip)
            try:
                fd_f, pos_f = self._fasta_writer.native_records_span(fasta)
                fd_v, pos_v = self._vcf_writer.native_span()
                eng.batch_fetch_to_files(fd_f if fasta else -1, pos_f, fd_v if len(vcf) else -1, pos_v)
                self._fasta_writer.commit_native_records(pos_f, fasta, bpl_last, last_line)
                self._vcf_writer.commit_native(pos_v, len(vcf))
                return
            except _ffi.MsimUnsupported:               # no regular files: through a mapping / write() made here
                pass
        if isinstance(fasta, int):
            region = self._fasta_writer.map_records(fasta)
            try:
                if fasta:
                    eng.batch_fetch_fasta(region.view)
            finally:
                self._fasta_writer.commit_records(region, bpl_last, last_line)
        else:
            self._fasta_writer.write_records(fasta, bpl_last, last_line)
        self._vcf_writer.write_raw(memoryview(vcf))

    def _batch_table(self, chroms, i, j):
        """``msim_batch_contig`` + ``msim_range`` tables of contigs [i, j) by array operations -- no Python per contig --
        when every contig is the std range over its whole length (ARGS mode: ``StdChromosomes``) and its defline can be
        handed over as it sits in the file.  None: the caller builds the batch contig by contig."""
        from .rmt import StdChromosomes
        fa = self._fasta
        tab = getattr(fa, "index_table", None)
        if not isinstance(chroms, StdChromosomes) or tab is None or getattr(fa, "name_bytes", None) is None:
            return None
        tab, names = tab[i:j], fa.name_bytes[i:j]
        n = j - i
        hlen = (tab["h1"] - tab["h0"]).astype(np.int64)
        if n == 0 or (names[:, 1] <= 0).any() or (hlen <= 0).any() or hlen.max() >= (1 << 31):
            return None
        ms = chroms.std
        probe = range_descriptor(type("RD", (), {"start": 0, "stop": 0, "mutation_settings": ms})())   # (fills the settings cache)
        del probe
        _, rate_sum, tmpl = _SETTINGS_CACHE[_settings_key(ms)]
        lens = tab["n_bases"].astype(np.int64)
        ranges = np.empty(n, dtype=_ffi.RANGE_DTYPE)
        ranges[:] = np.frombuffer(tmpl, dtype=_ffi.RANGE_DTYPE)[0]
        ranges["start"] = 0
        ranges["stop"] = lens - 1
        k = (lens.astype(np.float64) * rate_sum).astype(np.int64)          # int(((stop - start) + 1) * sum(rates))  mutator.py:225
        ranges["k"] = k
        ranges["setsize"] = sample_setsize_array(k)
        base = fa.text.ctypes.data
        t = np.zeros(n, dtype=_ffi.BATCH_CONTIG_DTYPE)
        t["body"] = base + tab["b0"]
        t["body_bytes"] = tab["b1"] - tab["b0"]
        t["n_bases"] = tab["n_bases"]
        t["lenc"] = tab["lenc"]
        t["lenb"] = tab["lenb"]
        t["ranges"] = ranges.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(_ffi.RANGE_DTYPE.itemsize)
        t["n_ranges"] = 1 if ms.has_mutations else 0                    # (plan_descriptors: ranges whose settings mutate)
        t["header"] = base + tab["h0"]
        t["header_len"] = hlen
        t["name"] = base + tab["h0"] + names[:, 0].astype(np.uint64)
        t["name_len"] = names[:, 1]
        return t, ranges, fa.text

    def _mutate_one(self, eng, chrom, earlier=()):
        """One contig through PLAN + APPLY + egress.  A device PLAN engine sizes its stream windows with 16-sigma
        margins; should one ever overflow, libmsim reports it (never a short read) and the contig is planned again by
        the sequential host planner -- same bytes, just slower.  The normal path pays nothing for this: the streams
        are put back to where the run started (they came from Python's generators) and the contigs before this one
        (`earlier`) are re-planned on the host, without output, to advance them."""
        done = set()                                   # what of this contig is already in the files / on stderr
        try:
            self._run_contig(eng, chrom, done)
        except _ffi.MsimError as e:
            # (fast RNG mode has no host planner to fall back to, and a replay through plan_chain would advance the
            #  contig ordinal -- every later contig, and every other rank of a --gpus run, would draw other numbers)
            if "overflowed its" not in str(e) or getattr(self._args, "rng", "compat") == "fast":
                raise
            global REPLANNED_CONTIGS
            REPLANNED_CONTIGS += 1
            eng.clear()
            export_python_streams(eng)                 # Python's generators still hold the states the run started with
            eng.set_plan_mode(_ffi.PLAN_HOST)
            try:
                for prev in (earlier[0][q] for q in range(earlier[1])) if earlier else ():
                    rec = self._fasta[prev.number]
                    # PLAN reads lengths and ranges only, never bases: any contig of the same length stands in
                    eng.plan_chain(len(rec), plan_table(prev))
                    eng.clear()
                self._run_contig(eng, chrom, done)
            finally:
                eng.set_plan_mode(_ffi.PLAN_AUTO)

    def _run_contig(self, eng, chrom, done):
        t = self._t
        t0 = time.perf_counter()
        rec = self._fasta[chrom.number]
        if getattr(rec, "uniform", False):            # file text -> HBM: strip + upper-case on the device
            cid = eng.add_contig_text(rec.body, len(rec), rec.lenc, rec.lenb)
        else:
            cid = eng.add_contig(rec.bases)
        t1 = time.perf_counter()
        t["ingest_s"] += t1 - t0
        table = plan_table(chrom)
        eng.plan_contig(cid, table)
        if eng.plan_was_empty(cid) and "warned" not in done:
            self._warn_empty(chrom)
            done.add("warned")
        bpl = self._fasta.faidx.index[rec.name].lenc
        if "header" not in done:                       # the reference writes it before it mutates (mutator.py:131-133)
            self._fasta_writer.set_bpl(bpl)
            self._fasta_writer.write_header(rec.long_name)
            done.add("header")
        eng.apply_contig(cid)
        # Egress.  Line framing and VCF text are rendered on the device and libmsim's output channels write them into the
        # files' next spans (csrc/file_io.hip: device -> pinned ring -> pwrite on a thread per file) while this thread goes on
        # with the next contig; mutate() joins them (file_wait) before the writers are closed.
        if bpl > 0:
            n_text = eng.fetch_sequence_framed_size(cid, bpl) if not NATIVE_FILE_EGRESS else None
            if n_text is None:
                out_len = eng.result_sizes(cid)[0]                     # (synchronises: PLAN + APPLY are done here)
                n_text = out_len + out_len // bpl
            t2 = time.perf_counter()
            t["plan_apply_s"] += t2 - t1
            q, r = divmod(n_text, bpl + 1)              # text = L + L // bpl bytes  ->  L
            fd, pos = self._fasta_writer.native_span()
            try:
                n_done = eng.fetch_sequence_framed_to_file(cid, bpl, fd, pos) if NATIVE_FILE_EGRESS else None
            except _ffi.MsimUnsupported:                # the file cannot be mapped there: through a mapping made here
                n_done = None
            if n_done is not None:
                self._fasta_writer.commit_native(pos, n_done, q * bpl + r)
            else:
                region = self._fasta_writer.map_region(n_text)
                try:
                    eng.fetch_sequence_framed_into(cid, bpl, region.view)
                finally:
                    self._fasta_writer.commit_region(region, q * bpl + r)
        else:
            text = eng.fetch_sequence(cid)
            t2 = time.perf_counter()
            t["plan_apply_s"] += t2 - t1
            self._fasta_writer.write_array(text)
        t3 = time.perf_counter()
        t["fasta_egress_s"] += t3 - t2
        fd, pos = self._vcf_writer.native_span()
        try:
            n_done = eng.render_vcf_device_to_file(cid, rec.name, fd, pos) if NATIVE_FILE_EGRESS else None
        except _ffi.MsimUnsupported:
            n_done = None
        if n_done is not None:
            self._vcf_writer.commit_native(pos, n_done)
        else:
            n_vcf = eng.render_vcf_device_size(cid, rec.name)
            region = self._vcf_writer.map_region(n_vcf)
            try:
                if n_vcf:
                    eng.render_vcf_device_into(cid, rec.name, region.view)
            finally:
                self._vcf_writer.commit_region(region)
        t["vcf_egress_s"] += time.perf_counter() - t3
        eng.clear()

    def _units(self, chroms):
        """mutate()'s contig loop (mutator.py:111-141) cut into units of work: a run of >= 2 small contigs (ONE pass through
        libmsim, msim_batch_run; at most BATCH_MAX_CONTIGS contigs and BATCH_MAX_BASES bases) or a single contig.  Units are
        also what a multi-GPU run shards (multi_gpu.py).  Array operations only where the contigs are the std ones."""
        from .rmt import StdChromosomes
        n = len(chroms)
        if self._fast_rng:                             # no chain to amortise, and batches are planned on the host's streams
            return [(q, q + 1) for q in range(n)]
        tab = getattr(self._fasta, "index_table", None)
        if (isinstance(chroms, StdChromosomes) and tab is not None and len(tab) == n
                and type(self)._batchable is Mutator._batchable):
            lens = tab["n_bases"].astype(np.int64)
            ms = chroms.std
            rate_sum = 0.0
            if ms.has_mutations:
                range_descriptor(type("RD", (), {"start": 0, "stop": 0, "mutation_settings": ms})())      # (fills the settings cache)
                rate_sum = float(_SETTINGS_CACHE[_settings_key(ms)][1])
            k = (lens.astype(np.float64) * rate_sum).astype(np.int64)       # int(((stop - start) + 1) * sum(rates))  mutator.py:225
            small = (lens <= BATCH_MAX_LEN) | ((lens <= BATCH_SPARSE_MAX_LEN) & (k <= BATCH_SPARSE_MAX_CANDIDATES))
            ok = ((tab["flags"] & _ffi.FASTA_NONUNIFORM) == 0) & (lens > 0) & small & (tab["lenc"] > 0)
        else:
            lens = np.fromiter((len(self._fasta[c.number]) for c in chroms), dtype=np.int64, count=n)
            ok = np.fromiter((self._batchable(c) for c in chroms), dtype=bool, count=n)
        cum = np.concatenate(([0], np.cumsum(lens)))
        edges = np.flatnonzero(np.diff(np.concatenate(([0], ok.astype(np.int8), [0]))))
        out, pos = [], 0
        for a, b in zip(edges[0::2].tolist(), edges[1::2].tolist()):      # maximal runs of batchable contigs
            out.extend((q, q + 1) for q in range(pos, a))
            i = a
            while i < b:
                j = int(np.searchsorted(cum, cum[i] + BATCH_MAX_BASES, side="right")) - 1
                j = min(j, b, i + BATCH_MAX_CONTIGS)
                if j - i >= 2:
                    out.append((i, j))
                    i = j
                else:
                    out.append((i, i + 1))
                    i += 1
            pos = b
        out.extend((q, q + 1) for q in range(pos, n))
        return out

    def _process_unit(self, eng, chroms, i, j):
        """A KeyError / ValueError inside a batch is replayed contig by contig (once: no re-batching behind the contig that
        raised), so that the files hold exactly what the reference had written by then."""
        if j - i >= 2:
            saved = (eng.get_mt_state(0), eng.get_mt_state(1))
            try:
                self._mutate_batch(eng, chroms, i, j)
                return
            except (KeyError, ValueError):
                eng.set_mt_state(0, *saved[0])
                eng.set_mt_state(1, *saved[1])
        for k in range(i, j):
            self._mutate_one(eng, chroms[k], earlier=(chroms, k))

    def _chromosomes(self):
        """The contigs in the order of the pass: the settings' own sequence (a list, or the lazy ``StdChromosomes``)."""
        from .rmt import StdChromosomes
        ch = self._sim.chromosomes
        return ch if isinstance(ch, StdChromosomes) else list(ch)

    def mutate(self):
        if int(getattr(self._args, "gpus", 1) or 1) > 1 and self._engine is None:
            from .multi_gpu import mutate_sharded      # one worker process per GPU, spawned before any GPU call here
            return mutate_sharded(self)
        if self._engine is None:
            self._engine = self._open_engine(getattr(self._args, "device", 0) or 0)
        else:
            self._seed_engine(self._engine)
        eng = self._engine
        eng.set_params(params_descriptor(self._sim))
        eng.reset_stats()
        try:
            chroms = self._chromosomes()
            for i, j in self._units(chroms):
                self._process_unit(eng, chroms, i, j)
            t0 = time.perf_counter()
            eng.file_wait()                            # what the output channels still hold goes out
            self._t["egress_wait_s"] = time.perf_counter() - t0
        finally:
            if not self._fast_rng:
                import_python_streams(eng)
            self.stats = eng.stats()
            self.stats["contig_path_s"] = {k: round(v, 4) for k, v in self._t.items()}
