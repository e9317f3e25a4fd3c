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

    def __init__(self, records: dict):
        self._records = records

    def __getitem__(self, name):
        r = self._records[name]
        return SimpleNamespace(rlen=len(r), offset=r.offset, lenc=r.lenc, lenb=r.lenb)

    def __iter__(self):
        return iter(self._records)

    def __len__(self):
        return len(self._records)


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
        self.index_table = None                        # msim_fasta_record per record, file order (numpy structured array)
        self.name_bytes = None                         # per record: where its name (first token of the defline) sits in it
        self._records: dict[str, FastaRecord] = {}
        self._order: list[FastaRecord] = []
        self._parse(raw)
        self.faidx = SimpleNamespace(index=_FaidxIndex(self._records))
        if write_index:
            self._write_fai()

    # ------------------------------------------------------------------ parsing
    def _parse(self, raw: np.ndarray) -> None:
        """The index pass runs in libmsim (csrc/fasta_index.cpp: msim_fasta_index, several host threads): per record the
        spans of the defline and of the body text, bases, bases / bytes per line and pyfaidx's line-length verdict.  A
        3 Gb genome has 24 records, an assembly tens of thousands -- the loop below is all the per-record Python left."""
        from . import _ffi
        if raw.shape[0] == 0:
            return
        idx = _ffi.fasta_index(raw)
        if idx is None:
            raise FastaIndexingError("Sequence data found before the first defline")
        self.index_table = idx
        h0, h1, b0, b1 = (idx[k].tolist() for k in ("h0", "h1", "b0", "b1"))
        n_bases, lenc, lenb, flags = (idx[k].tolist() for k in ("n_bases", "lenc", "lenb", "flags"))
        whole = raw.tobytes() if raw.shape[0] < (256 << 20) and idx.shape[0] > 1000 else None   # (deflines of an assembly: one copy
        name_bytes = []                                                                         #  beats 10^5 tiny ones)
        for k in range(idx.shape[0]):
            hb = whole[h0[k]:h1[k]] if whole is not None else raw[h0[k]:h1[k]].tobytes()
            long_name = hb.decode("utf-8", "replace")
            toks = long_name.split()
            name = toks[0] if toks else ""
            nb = name.encode("utf-8")                  # where the name sits in the file text (the batch path points libmsim at it)
            at = hb.find(nb) if nb else -1
            name_bytes.append((at, len(nb)) if at >= 0 and long_name.encode("utf-8", "replace") == hb else (-1, 0))
            if name in self._records:
                raise ValueError(f"Duplicate key \"{name}\"")
            fl = flags[k]
            if not fl & _ffi.FASTA_HAS_BODY:
                rec = FastaRecord(name, long_name, np.zeros(0, np.uint8), 0, 0, 0, b0[k], True)
            else:
                if fl & _ffi.FASTA_BAD_LINES:
                    raise FastaIndexingError(
                        f"Line length of fasta file is not consistent in {name}")
                rec = FastaRecord(name, long_name, raw[b0[k]:b1[k]], n_bases[k], lenc[k], lenb[k], b0[k],
                                  not fl & _ffi.FASTA_NONUNIFORM)
            self._records[name] = rec
            self._order.append(rec)
        self.name_bytes = np.array(name_bytes, dtype=np.int64).reshape(-1, 2)    # (offset inside the defline, bytes) or (-1, 0)

    def _write_fai(self) -> None:
        path = self.filename + ".fai"
        if os.path.exists(path):
            return
        try:
            with open(path, "w") as fh:
                for r in self._order:
                    fh.write(f"{r.name}\t{len(r)}\t{r.offset}\t{r.lenc}\t{r.lenb}\n")
        except OSError:
            pass

    # ------------------------------------------------------------------ pyfaidx-like surface
    def keys(self):
        return self._records.keys()

    def __len__(self) -> int:
        return len(self._order)

    def __getitem__(self, key) -> FastaRecord:
        if isinstance(key, (int, np.integer)):
            return self._order[key]
        return self._records[key]

    def __iter__(self):
        return iter(self._order)

    def get_seq(self, name: str, start: int, end: int) -> str:
        return self._records[name][start - 1:end]

    def close(self) -> None:
        pass                                           # (the mapping goes with the last array that views it)
