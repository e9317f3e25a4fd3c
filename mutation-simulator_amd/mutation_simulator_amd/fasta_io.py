"""FASTA ingest for the MI355X path: whole contigs as upper-cased ``uint8`` arrays.

Replaces what the reference gets from the third-party ``pyfaidx`` package (reference
util.py:77-91: ``Fasta(path, one_based_attributes=False, as_raw=True, sequence_always_upper=True,
read_ahead=10000)``).  The reference streams bases one at a time through pyfaidx; the HIP path wants
each contig as one contiguous byte array to upload to HBM, so this loader indexes the file once (natively, in
libmsim: ``msim_fasta_index``) and exposes the small part of the pyfaidx surface the path touches:

    fasta.keys()                       record names in file order
    fasta[i] / fasta[name]             -> FastaRecord
    record.name / record.long_name     first whitespace token / whole defline
    len(record), record[pos], record[a:b]   (upper-cased str, like as_raw=True)
    fasta.faidx.index[name].lenc       bases per line of that record
    fasta.close()

plus ``record.body`` (the record's file text, for ingest on the device: the HIP gather kernel skips the
line terminators and upper-cases) and ``record.bases`` (np.uint8, materialised lazily on the host).  Like
pyfaidx it leaves a samtools-style ``<infile>.fai`` next to the input when it can.
"""
from __future__ import annotations

import mmap
import os
from collections.abc import Mapping
from types import SimpleNamespace

import numpy as np


class FastaIndexingError(Exception):
    """Malformed FASTA (inconsistent line lengths, data before the first defline ...)."""


class FastaNotFoundError(Exception):
    """Input FASTA does not exist / is unreadable."""


class FastaRecord:
    """One record of the index.  ``body`` is the record's file text (a view of the loaded file: the bytes
    after the header line); the upper-cased base array is only materialised on the host if someone asks for
    ``bases`` -- the device path ingests ``body`` directly (``Engine.add_contig_text``)."""
    __slots__ = ("name", "long_name", "body", "n_bases", "lenc", "lenb", "offset", "uniform", "_bases")

    def __init__(self, name: str, long_name: str, body: np.ndarray, n_bases: int, lenc: int, lenb: int,
                 offset: int, uniform: bool):
        self.name = name
        self.long_name = long_name
        self.body = body
        self.n_bases = n_bases
        self.lenc = lenc
        self.lenb = lenb
        self.offset = offset
        self.uniform = uniform        # every line of `lenc` bases is `lenb` bytes long: bases sit at fixed strides
        self._bases = None

    @property
    def bases(self) -> np.ndarray:
        if self._bases is None:
            seg = self.body
            b = seg[(seg != 10) & (seg != 13)]
            _upper_inplace(b)
            self._bases = b
        return self._bases

    def __len__(self) -> int:
        return self.n_bases

    def __getitem__(self, key) -> str:
        if isinstance(key, slice):
            return self.bases[key].tobytes().decode("latin-1")
        return chr(int(self.bases[key]))

    def __str__(self) -> str:
        return self.bases.tobytes().decode("latin-1")


def _upper_inplace(a: np.ndarray) -> None:
    lower = (a >= 97) & (a <= 122)
    a[lower] -= 32


class _FaidxIndex(Mapping):
    """``fasta.faidx.index``: name -> (rlen, offset, lenc, lenb), built on demand (an assembly has 10^5 records; the
    mutation pass asks for a handful of line widths)."""

    def __init__(self, fasta: "Fasta"):
        self._fasta = fasta

    def __getitem__(self, name):
        row = self._fasta.index_table[self._fasta._by_name[name]]
        has = bool(row["flags"] & 1)
        return SimpleNamespace(rlen=int(row["n_bases"]) if has else 0, offset=int(row["b0"]),
                               lenc=int(row["lenc"]) if has else 0, lenb=int(row["lenb"]) if has else 0)

    def __iter__(self):
        return iter(self._fasta._by_name)

    def __len__(self):
        return len(self._fasta._by_name)


class Fasta:
    def __init__(self, filename, write_index: bool = True, **_pyfaidx_compat):
        filename = str(filename)
        if not os.path.isfile(filename):
            raise FastaNotFoundError(f"Cannot read FASTA from file {filename}")
        self.filename = filename
        # mapped, not read: the index pass and the per-record uploads touch the page cache directly (a 3 GB np.fromfile is a
        # second copy of the genome in memory and ~0.1 s per GB before anything else can start)
        # (a plain ndarray over the mapping, not np.memmap: slicing a memmap costs microseconds per record)
        size = os.path.getsize(filename)
        if size:
            with open(filename, "rb") as fh:
                self._map = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            raw = np.frombuffer(self._map, dtype=np.uint8)
        else:
            self._map = None
            raw = np.zeros(0, np.uint8)
        self.text = raw                                # the whole file text (the batch path hands spans of it to libmsim)
        self.text_bytes = size
        self.index_table = _ffi_empty_index()          # msim_fasta_record per record, file order (numpy structured array)
        self.name_bytes = np.zeros((0, 2), dtype=np.int64)   # per record: where its name (first token of the defline) sits in it
        self._names: list[str] = []
        self._by_name: dict[str, int] = {}
        self._cache: dict[int, FastaRecord] = {}
        self._parse(raw)
        self.faidx = SimpleNamespace(index=_FaidxIndex(self))
        if write_index:
            self._write_fai()

    # ------------------------------------------------------------------ parsing
    def _parse(self, raw: np.ndarray) -> None:
        """The index pass runs in libmsim (csrc/fasta_index.cpp: msim_fasta_index, several host threads): per record the
        spans of the defline and of the body text, bases, bases / bytes per line and pyfaidx's line-length verdict.  A
        3 Gb genome has 24 records, an assembly hundreds of thousands: per record this loop only takes the NAME out of the
        defline (duplicates are an error, like pyfaidx's) -- ``FastaRecord`` objects are made when somebody asks for one."""
        from . import _ffi
        if raw.shape[0] == 0:
            return
        try:
            idx = _ffi.fasta_index(raw)
        except _ffi.MsimError:                         # no loadable libmsim (a host without ROCm): tooling that only wants
            idx = _index_python(raw, _ffi)             # the loader still gets its index, from the plain restatement below
        if idx is None:
            raise FastaIndexingError("Sequence data found before the first defline")
        self.index_table = idx
        n = idx.shape[0]
        h0 = idx["h0"].astype(np.int64)
        hl = idx["h1"].astype(np.int64) - h0
        # every defline in one gather (they are tiny next to the bodies)
        starts = np.concatenate(([0], np.cumsum(hl)))
        total = int(starts[-1])
        gather = np.arange(total, dtype=np.int64) + np.repeat(h0 - starts[:-1], hl)
        blob = raw[gather].tobytes() if total else b""
        try:
            text = blob.decode("ascii")
        except UnicodeDecodeError:
            text = None
        flags = idx["flags"].tolist()
        offs = starts.tolist()
        names, by_name = self._names, self._by_name
        where = np.empty((n, 2), dtype=np.int64)
        for k in range(n):
            o, e = offs[k], offs[k + 1]
            if text is not None:
                long_name = text[o:e]
                toks = long_name.split(None, 1)
                name = toks[0] if toks else ""
                at, nb = (len(long_name) - len(long_name.lstrip()), len(name)) if toks else (-1, 0)
            else:
                hb = blob[o:e]
                long_name = hb.decode("utf-8", "replace")
                toks = long_name.split(None, 1)
                name = toks[0] if toks else ""
                enc = name.encode("utf-8")
                at = hb.find(enc) if enc else -1
                at, nb = (at, len(enc)) if at >= 0 and long_name.encode("utf-8", "replace") == hb else (-1, 0)
            if name in by_name:
                raise ValueError(f"Duplicate key \"{name}\"")
            fl = flags[k]
            if (fl & _ffi.FASTA_HAS_BODY) and (fl & _ffi.FASTA_BAD_LINES):
                raise FastaIndexingError(f"Line length of fasta file is not consistent in {name}")
            by_name[name] = k
            names.append(name)
            where[k, 0], where[k, 1] = at, nb
        self.name_bytes = where

    def _record(self, k: int) -> FastaRecord:
        rec = self._cache.get(k)
        if rec is None:
            from . import _ffi
            row = self.index_table[k]
            long_name = self.text[int(row["h0"]):int(row["h1"])].tobytes().decode("utf-8", "replace")
            fl = int(row["flags"])
            if not fl & _ffi.FASTA_HAS_BODY:
                rec = FastaRecord(self._names[k], long_name, np.zeros(0, np.uint8), 0, 0, 0, int(row["b0"]), True)
            else:
                rec = FastaRecord(self._names[k], long_name, self.text[int(row["b0"]):int(row["b1"])], int(row["n_bases"]),
                                  int(row["lenc"]), int(row["lenb"]), int(row["b0"]), not fl & _ffi.FASTA_NONUNIFORM)
            if len(self._cache) > (1 << 20):           # (bounded: an assembly is walked once, front to back)
                self._cache.clear()
            self._cache[k] = rec
        return rec

    def _write_fai(self) -> None:
        path = self.filename + ".fai"
        if os.path.exists(path):
            return
        try:
            t = self.index_table
            has = (t["flags"] & 1) != 0
            cols = zip(self._names, np.where(has, t["n_bases"], 0).tolist(), t["b0"].tolist(),
                       np.where(has, t["lenc"], 0).tolist(), np.where(has, t["lenb"], 0).tolist())
            with open(path, "w") as fh:
                fh.write("".join(f"{a}\t{b}\t{c}\t{d}\t{e}\n" for a, b, c, d, e in cols))
        except OSError:
            pass

    # ------------------------------------------------------------------ pyfaidx-like surface
    def keys(self):
        return self._by_name.keys()

    def __len__(self) -> int:
        return len(self._names)

    def __getitem__(self, key) -> FastaRecord:
        if isinstance(key, (int, np.integer)):
            k = int(key)
            if k < 0:
                k += len(self._names)
            if not 0 <= k < len(self._names):
                raise IndexError("record index out of range")
            return self._record(k)
        return self._record(self._by_name[key])

    def __iter__(self):
        return (self._record(k) for k in range(len(self._names)))

    def names_and_lengths(self):
        """(name, bases) of every record in file order, without making the record objects."""
        t = self.index_table
        return zip(self._names, np.where((t["flags"] & 1) != 0, t["n_bases"], 0).tolist())

    def get_seq(self, name: str, start: int, end: int) -> str:
        return self[name][start - 1:end]

    def close(self) -> None:
        pass                                           # (the mapping goes with the last array that views it)


def _index_python(raw: np.ndarray, _ffi):
    """``msim_fasta_index`` restated line by line in Python (slow: a fallback for hosts where libmsim.so cannot be loaded,
    and what tests/test_fasta_index.py checks the native pass against).  Same result array, or None for sequence text
    before the first defline."""
    text = raw.tobytes()
    lines, at = [], 0
    while at < len(text):
        e = text.find(b"\n", at)
        if e < 0:
            e = len(text)
        lines.append((at, e))
        at = e + 1
    recs, cur = [], None
    for a, e in lines:
        if text[a:a + 1] == b">":
            cr = e > a + 1 and text[e - 1:e] == b"\r"
            cur = [a + 1, e - cr, e + 1, []]
            recs.append(cur)
        elif cur is None:
            if e > a:
                return None
        else:
            cr = e > a and text[e - 1:e] == b"\r"
            cur[3].append((a, e, e - a - cr, cr))
    out = np.zeros(len(recs), dtype=_ffi.FASTA_RECORD_DTYPE)
    for k, (h0, h1, after, ls) in enumerate(recs):
        out["h0"][k], out["h1"][k] = h0, h1
        if not ls:
            out["b0"][k] = out["b1"][k] = after
            continue
        lenc, first_cr = ls[0][2], ls[0][3]
        nz = [i for i, l in enumerate(ls) if l[2] > 0]
        last = nz[-1] if nz else -1
        bad = any(l[2] != lenc for l in ls[:max(last, 0)]) or (last >= 0 and ls[last][2] > lenc)
        nonuni = any(l[3] != first_cr for l in ls[:max(last, 0)])
        out["b0"][k], out["b1"][k] = ls[0][0], ls[-1][1]
        out["n_bases"][k] = sum(l[2] for l in ls)
        out["lenc"][k], out["lenb"][k] = lenc, ls[0][1] - ls[0][0] + 1
        out["flags"][k] = 1 | (2 if bad else 0) | (4 if nonuni else 0)
    return out


def _ffi_empty_index():
    from . import _ffi
    return np.zeros(0, dtype=_ffi.FASTA_RECORD_DTYPE)
