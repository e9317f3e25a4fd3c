"""Fasta output (reference fasta_writer.py:13-65): same class surface, bulk-capable.

The reference wraps lines by counting single-base ``write`` calls.  The HIP path hands back a whole
mutated contig as one ``uint8`` array, so ``write_array`` frames it in a few vectorised passes
while keeping the exact byte semantics: a newline after every ``bpl``-th base, none after a partial
last line, and a newline *before* the next header only if the previous line was partial.
"""
from __future__ import annotations

import ctypes
import errno
import mmap
import os

import numpy as np

_libc = None


def fallocate_native(fd: int, offset: int, nbytes: int) -> bool:
    """Linux ``fallocate(2)``, mode 0, called directly.  True: the span is allocated.  False: the filesystem has no
    native fallocate (EOPNOTSUPP / ENOSYS / EINVAL) -- nothing was touched.  Other errors (ENOSPC ...) raise.

    Deliberately NOT ``os.posix_fallocate``: where the filesystem lacks the operation (NFS, some FUSE / overlay
    mounts) glibc *emulates* it by reading one byte per block and writing a zero back where it read none -- a byte
    another thread (or a device-to-host copy into a mapping of the same span) writes between that read and that
    write is silently zeroed, and the emulation costs a syscall per 4 KiB block."""
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(None, use_errno=True)
        _libc.fallocate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
        _libc.fallocate.restype = ctypes.c_int
    while True:
        if _libc.fallocate(fd, 0, offset, nbytes) == 0:
            return True
        e = ctypes.get_errno()
        if e == errno.EINTR:
            continue
        if e in (errno.EOPNOTSUPP, errno.ENOSYS, errno.EINVAL):
            return False
        raise OSError(e, os.strerror(e))


class MappedRegion:
    """The next ``nbytes`` of an output file as a writable uint8 array (``view``): the file is extended and that span
    mapped, so whoever fills it -- a device-to-host copy -- writes straight into the page cache.  ``close`` unmaps.
    The fallback of libmsim's output channels (csrc/file_io.hip), which write the same spans with pwrite() from threads of
    their own: building and tearing down the page tables of a mapping costs more than the copy (measured there)."""
    __slots__ = ("pos", "nbytes", "view", "_map")

    def __init__(self, fileobj, nbytes: int):
        fileobj.flush()
        self.pos = fileobj.tell()
        self.nbytes = int(nbytes)
        self._map = None
        self.view = None
        if self.nbytes:
            fd = fileobj.fileno()
            if os.fstat(fd).st_size < self.pos + self.nbytes:
                # allocate the span's pages in bulk: faulting fresh pages of a just-extended file in one by one costs
                # 40-50 ms per 200 MB on tmpfs, fallocate 9 (and the copy into the allocated span 15)
                # (no space left raises: better an exception here than a SIGBUS in the copy)
                if not fallocate_native(fd, self.pos, self.nbytes):
                    os.ftruncate(fd, self.pos + self.nbytes)
            start = self.pos - self.pos % mmap.ALLOCATIONGRANULARITY
            self._map = mmap.mmap(fd, self.pos + self.nbytes - start, access=mmap.ACCESS_WRITE, offset=start)
            self.view = np.frombuffer(self._map, dtype=np.uint8)[self.pos - start:]
        else:
            self.view = np.zeros(0, dtype=np.uint8)

    def close(self, fileobj):
        self.view = None
        if self._map is not None:
            self._map.close()
            self._map = None
        fileobj.seek(self.pos + self.nbytes)


class FastaWriterError(Exception):
    """Raised when the writer can not write to a file."""


class FastaWriter:
    def __init__(self, fname):
        try:
            self._out = open(fname, "w+b")         # (readable too: map_region maps spans of it)
        except IOError as e:
            raise FastaWriterError(f"Cannot write to Fasta file {fname} {e}")
        self._written = 0      # bases on the current (partial) line
        self._bpl = 60

    def __del__(self):
        self.close()

    def close(self):
        out = getattr(self, "_out", None)
        if out is not None and not out.closed:
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

    def map_region(self, nbytes: int) -> MappedRegion:
        """The span a framed record body of ``nbytes`` bytes will occupy, mapped for writing (see ``write_framed``; must
        start a line).  ``commit_region(region, n_bases)`` finishes it."""
        if self._written != 0:
            raise FastaWriterError("map_region needs to start at the beginning of a line")
        return MappedRegion(self._out, nbytes)

    def commit_region(self, region: MappedRegion, n_bases: int):
        region.close(self._out)
        if region.nbytes:
            self._written = n_bases % self._bpl

    def native_span(self):
        """(fd, position) where the next framed record body goes, for a writer outside Python (libmsim's
        ``msim_fetch_sequence_framed_file``); must start a line.  ``commit_native(pos, nbytes, n_bases)`` finishes it."""
        if self._written != 0:
            raise FastaWriterError("native_span needs to start at the beginning of a line")
        self._out.flush()
        return self._out.fileno(), self._out.tell()

    def commit_native(self, pos: int, nbytes: int, n_bases: int):
        self._out.seek(pos + nbytes)
        if nbytes:
            self._written = n_bases % self._bpl

    def native_records_span(self, nbytes: int):
        """(fd, position) for a run of complete records of ``nbytes`` bytes written outside Python (see ``write_records``;
        the newline a partial previous line is owed is written here); ``commit_native_records`` finishes it."""
        if nbytes and self._written != 0:
            self._out.write(b"\n")
            self._written = 0
        self._out.flush()
        return self._out.fileno(), self._out.tell()

    def commit_native_records(self, pos: int, nbytes: int, bpl: int, last_line_bases: int):
        self._out.seek(pos + nbytes)
        if nbytes:
            self._bpl = bpl
            self._written = int(last_line_bases)

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

    def map_records(self, nbytes: int) -> MappedRegion:
        """The span a run of complete records of ``nbytes`` bytes will occupy (see ``write_records``), mapped for writing;
        ``commit_records(region, bpl, last_line_bases)`` finishes it."""
        if nbytes and self._written != 0:
            self._out.write(b"\n")
            self._written = 0
        return MappedRegion(self._out, nbytes)

    def commit_records(self, region: MappedRegion, bpl: int, last_line_bases: int):
        region.close(self._out)
        if region.nbytes:
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
