"""BEDPE output of the interchromosomal-translocation pass (reference bedpe_writer.py:13-55): same class surface and line
format; ``write`` builds all lines of one contig at once."""
from __future__ import annotations


class BedpeWriterError(Exception):
    """Raised when the writer can not write to a file."""


class BedpeWriter:
    def __init__(self, fname):
        try:
            self._out = open(fname, "w")
        except IOError as e:
            raise BedpeWriterError(f"Cannot write to BEDPE file {fname} {e}")

    def __del__(self):
        self.close()

    def close(self):
        out = getattr(self, "_out", None)
        if out is not None and not out.closed:
            out.close()

    def write_header(self):
        """(The reference defines it and never calls it: its BEDPE files carry no header line.)"""
        self._out.write("#chrom1\tstart1\tstop1\tchrom2\tstart2\tstop2\n")

    def write(self, chrom: str, bp_chrom, chrom_len_pre_it: int, partner: str, bp_partner, partner_len_pre_it: int):
        """One line per PAIR of consecutive breakpoints -- the stretch between them is what moved -- and, for an odd
        number of them, a last line from the last breakpoint to the ends of both contigs (bedpe_writer.py:44-55)."""
        a = [int(x) for x in bp_chrom]
        b = [int(x) for x in bp_partner]
        n = len(a)
        lines = [f"{chrom}\t{a[i]}\t{a[i + 1]}\t{partner}\t{b[i]}\t{b[i + 1]}\n" for i in range(0, n - 1, 2)]
        if n % 2:
            lines.append(f"{chrom}\t{a[n - 1]}\t{chrom_len_pre_it}\t{partner}\t{b[n - 1]}\t{partner_len_pre_it}\n")
        self._out.write("".join(lines))
