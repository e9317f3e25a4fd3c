"""Fasta output (reference fasta_writer.py:13-65): same class surface, bulk-capable.

The reference wraps lines by counting single-base ``write`` calls.  The HIP path hands back a whole
mutated contig as one ``uint8`` array, so ``write_array`` frames it in a few vectorised passes
while keeping the exact byte semantics: a newline after every ``bpl``-th base, none after a partial
last line, and a newline *before* the next header only if the previous line was partial.
"""
from __future__ import annotations

import numpy as np


class FastaWriterError(Exception):
    """Raised when the writer can not write to a file."""


class FastaWriter:
    def __init__(self, fname):
        try:
            self._out = open(fname, "wb")
        except IOError as e:
            raise FastaWriterError(f"Cannot write to Fasta file {fname} {e}")
        self._written = 0      # bases on the current (partial) line
        self._bpl = 60

    def __del__(self):
        self.close()

    def close(self):
        out = getattr(self, "_out", None)
        if out is not None:
            out.close()

    def set_bpl(self, bpl: int):
        self._bpl = bpl

    def write_header(self, header: str):
        if self._written != 0:
            self._out.write(b"\n")
        self._out.write(b">" + header.encode("utf-8", "replace") + b"\n")
        self._written = 0

    def write(self, base: str):
        self.write_array(np.frombuffer(base.encode("latin-1"), dtype=np.uint8))

    def write_multi(self, bases):
        if not isinstance(bases, str):
            bases = "".join(bases)
        self.write_array(np.frombuffer(bases.encode("latin-1"), dtype=np.uint8))

    def write_framed(self, text: np.ndarray, n_bases: int):
        """Append a record body that is already wrapped at the current line width (framed on the device,
        ``Engine.fetch_sequence_framed``); must start a line, i.e. directly follow ``write_header``."""
        if self._written != 0:
            raise FastaWriterError("write_framed needs to start at the beginning of a line")
        if n_bases:
            self._out.write(memoryview(np.ascontiguousarray(text)))
            self._written = n_bases % self._bpl

    def write_records(self, text: np.ndarray, bpl: int, last_line_bases: int):
        """Append a run of complete records -- header lines and wrapped bodies, exactly the bytes ``write_header`` +
        ``write_framed`` would produce for each (``Engine.batch_run``).  ``bpl`` / ``last_line_bases``: line width of the
        last record and the bases on its last line, so that whatever follows continues correctly."""
        if text.shape[0] == 0:
            return
        if self._written != 0:
            self._out.write(b"\n")
        self._out.write(memoryview(np.ascontiguousarray(text)))
        self._bpl = bpl
        self._written = int(last_line_bases)

    def begin_segment(self):
        """Multi-GPU workers write the records they own into a part file of their own: a segment starts as if at the
        beginning of a file (no newline owed to whatever precedes it there); ``append_segment`` of the assembling
        process restores the rule between segments."""
        self._written = 0

    def tell(self) -> int:
        self._out.flush()
        return self._out.tell()

    def append_segment(self, src, offset: int, nbytes: int):
        """Append bytes [offset, offset + nbytes) of the open binary file ``src`` -- complete records written by another
        writer after ``begin_segment`` -- keeping the reference's newline rule: a record that ended mid-line is followed
        by a newline before the next header (fasta_writer.py:40-47)."""
        if nbytes <= 0:
            return
        if self._written != 0:
            self._out.write(b"\n")
        src.seek(offset)
        left, last = nbytes, b"\n"
        while left:
            chunk = src.read(min(left, 64 << 20))
            if not chunk:
                raise FastaWriterError("segment file shorter than its index says")
            self._out.write(chunk)
            last = chunk[-1:]
            left -= len(chunk)
        self._written = 0 if last == b"\n" else 1          # (only zero / non-zero matters before a header)

    def write_array(self, bases: np.ndarray):
        """Append ``bases`` (uint8) wrapped at the current line width."""
        n = int(bases.shape[0])
        if n == 0:
            return
        bpl = self._bpl
        pos = 0
        if self._written:
            take = min(bpl - self._written, n)
            self._out.write(bases[:take].tobytes())
            self._written += take
            pos = take
            if self._written == bpl:
                self._out.write(b"\n")
                self._written = 0
        full = (n - pos) // bpl
        if full:
            # stream in slabs so a 250 Mb contig does not need a second full-size buffer at once
            slab = max(1, (64 << 20) // (bpl + 1))
            for a in range(0, full, slab):
                b = min(full, a + slab)
                block = np.empty((b - a, bpl + 1), dtype=np.uint8)
                block[:, :bpl] = bases[pos + a * bpl: pos + b * bpl].reshape(b - a, bpl)
                block[:, bpl] = 10
                self._out.write(block.tobytes())
            pos += full * bpl
        if pos < n:
            self._out.write(bases[pos:].tobytes())
            self._written = n - pos
