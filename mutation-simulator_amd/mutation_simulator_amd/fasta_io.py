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
        # per-line quantities for the whole file, per-record reductions with reduceat: an assembly with tens of
        # thousands of scaffolds must not pay a dozen small NumPy calls per record
        n_lines = starts.shape[0]
        l_start = starts.astype(np.int64)
        l_end = ends.astype(np.int64)
        cr = (l_end > l_start) & (raw[np.maximum(l_end - 1, 0)] == 13)
        llen = l_end - l_start - cr
        line_idx = np.arange(n_lines, dtype=np.int64)
        body_line = ~is_hdr
        body_line[:hdr_lines[0]] = False
        rec_of_line = np.cumsum(is_hdr) - 1                               # (lines before the first defline: -1, masked)
        lo = hdr_lines + 1
        hi = np.concatenate((hdr_lines[1:], [n_lines]))
        has_body = lo < hi
        lo_c = np.minimum(lo, n_lines - 1)
        n_bases = np.add.reduceat(np.where(body_line, llen, 0), hdr_lines)
        lenc_r = np.where(has_body, llen[lo_c], 0)
        lenb_r = np.where(has_body, l_end[lo_c] - l_start[lo_c] + 1, 0)
        last_nz = np.maximum.reduceat(np.where(body_line & (llen > 0), line_idx, -1), hdr_lines)
        rl = np.maximum(rec_of_line, 0)
        before_last = body_line & (line_idx < last_nz[rl])
        # every line but the last non-empty one must have the record's line length
        bad_line = (before_last & (llen != lenc_r[rl])) | (body_line & (line_idx == last_nz[rl]) & (llen > lenc_r[rl]))
        bad_rec = np.logical_or.reduceat(bad_line, hdr_lines)
        # fixed stride: every full line has the first line's terminator ('\n' or '\r\n')
        nonuni = np.logical_or.reduceat(before_last & (cr != cr[lo_c][rl]), hdr_lines)
        h0s = l_start[hdr_lines] + 1
        h1s = l_end[hdr_lines] - cr[hdr_lines]
        b0s = l_start[lo_c]
        b1s = l_end[np.maximum(hi - 1, 0)]
        for k in range(hdr_lines.shape[0]):
            long_name = raw[h0s[k]:h1s[k]].tobytes().decode("utf-8", "replace")
            toks = long_name.split()
            name = toks[0] if toks else ""
            if name in self._records:
                raise ValueError(f"Duplicate key \"{name}\"")
            if not has_body[k]:
                rec = FastaRecord(name, long_name, np.zeros(0, np.uint8), 0, 0, 0, int(l_end[hdr_lines[k]]) + 1, True)
            else:
                if bad_rec[k]:
                    raise FastaIndexingError(
                        f"Line length of fasta file is not consistent in {name}")
                b0 = int(b0s[k])
                rec = FastaRecord(name, long_name, raw[b0:int(b1s[k])], int(n_bases[k]), int(lenc_r[k]), int(lenb_r[k]),
                                  b0, not bool(nonuni[k]))
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
