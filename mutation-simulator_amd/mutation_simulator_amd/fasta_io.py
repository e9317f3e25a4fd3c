"""FASTA ingest for the MI355X path: whole contigs as upper-cased ``uint8`` arrays.

Replaces what the reference gets from the third-party ``pyfaidx`` package (reference
util.py:77-91: ``Fasta(path, one_based_attributes=False, as_raw=True, sequence_always_upper=True,
read_ahead=10000)``).  The reference streams bases one at a time through pyfaidx; the HIP path wants
each contig as one contiguous byte array to upload to HBM, so this loader parses the file once with
vectorised NumPy passes and exposes the small part of the pyfaidx surface the path touches:

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

import os
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


class Fasta:
    def __init__(self, filename, write_index: bool = True, **_pyfaidx_compat):
        filename = str(filename)
        if not os.path.isfile(filename):
            raise FastaNotFoundError(f"Cannot read FASTA from file {filename}")
        self.filename = filename
        raw = np.fromfile(filename, dtype=np.uint8)
        self._records: dict[str, FastaRecord] = {}
        self._order: list[FastaRecord] = []
        self._parse(raw)
        self.faidx = SimpleNamespace(index={
            r.name: SimpleNamespace(rlen=len(r), offset=r.offset, lenc=r.lenc, lenb=r.lenb)
            for r in self._order})
        if write_index:
            self._write_fai()

    # ------------------------------------------------------------------ parsing
    def _parse(self, raw: np.ndarray) -> None:
        n = raw.shape[0]
        if n == 0:
            return
        nl = np.flatnonzero(raw == 10)
        starts = np.concatenate(([0], nl + 1))
        if starts[-1] >= n:
            starts = starts[:-1]
        ends = np.concatenate((nl, [n]))[:starts.shape[0]]          # exclusive, at the '\n'
        is_hdr = raw[starts] == ord(">")
        hdr_lines = np.flatnonzero(is_hdr)
        if hdr_lines.size == 0:
            if np.any(ends > starts):
                raise FastaIndexingError("Sequence data found before the first defline")
            return
        if hdr_lines[0] != 0 and np.any(ends[:hdr_lines[0]] > starts[:hdr_lines[0]]):
            raise FastaIndexingError("Sequence data found before the first defline")
        for k, li in enumerate(hdr_lines):
            h0, h1 = int(starts[li]) + 1, int(ends[li])
            if h1 > h0 and raw[h1 - 1] == 13:
                h1 -= 1
            long_name = raw[h0:h1].tobytes().decode("utf-8", "replace")
            toks = long_name.split()
            name = toks[0] if toks else ""
            lo = li + 1
            hi = int(hdr_lines[k + 1]) if k + 1 < hdr_lines.size else starts.shape[0]
            if name in self._records:
                raise ValueError(f"Duplicate key \"{name}\"")
            if lo >= hi:
                rec = FastaRecord(name, long_name, np.zeros(0, np.uint8), 0, 0, 0, int(ends[li]) + 1, True)
            else:
                b0, b1 = int(starts[lo]), int(ends[hi - 1])
                l_end = ends[lo:hi].astype(np.int64)
                l_start = starts[lo:hi].astype(np.int64)
                cr = (l_end > l_start) & (raw[np.maximum(l_end - 1, 0)] == 13)
                llen = l_end - l_start - cr
                lenc = int(llen[0])
                lenb = int(l_end[0] - l_start[0]) + 1
                # every line but the last non-empty one must have the record's line length
                nz = np.flatnonzero(llen > 0)
                if nz.size:
                    body = llen[:nz[-1]]
                    if np.any(body != lenc) or llen[nz[-1]] > lenc:
                        raise FastaIndexingError(
                            f"Line length of fasta file is not consistent in {name}")
                # fixed stride: every full line has the first line's terminator ('\n' or '\r\n')
                uniform = bool(nz.size == 0 or np.all(cr[:nz[-1]] == cr[0]))
                rec = FastaRecord(name, long_name, raw[b0:b1], int(llen.sum()), lenc, lenb, b0, uniform)
            self._records[name] = rec
            self._order.append(rec)

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
        pass
