"""VCF output (reference vcf_writer.py:16-126): ``VcfRecord`` / ``VcfWriter`` with the same
fields, header text and line format; ``write_raw`` takes pre-rendered record lines from the C-ABI
renderer (``msim_render_vcf``) so 30 M records do not go through Python objects."""
from __future__ import annotations

import os
from datetime import datetime


class VcfWriterError(Exception):
    """Raised when the writer can not write to a file."""


class VcfRecord:
    def __init__(self, svtype: str = "", start: int = 0, end: int = 0, len: int = 0,
                 ref: str = "", alt: str = ""):
        self.svtype = svtype
        self.start = start
        self.end = end
        self.len = len
        self.ref = ref
        self.alt = alt

    def __repr__(self) -> str:
        return f"{self.svtype} {self.start} {self.end} {self.len} {self.ref} {self.alt}"

    @property
    def info(self) -> str:
        if self.svtype == "sn":
            return "."
        return f"SVTYPE={self.svtype};END={self.end};SVLEN={self.len}"


_HEADER_TAIL = (
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">\n'
    '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the variant described in '
    'this record">\n'
    '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="Difference in length between REF and ALT '
    'alleles">\n'
    '##ALT=<ID=INS,Description="Insert">\n'
    '##ALT=<ID=DEL,Description="Deletion">\n'
    '##ALT=<ID=DUP,Description="Duplication">\n'
    '##ALT=<ID=INV,Description="Inversion">\n'
    '##ALT=<ID=DEL:ME,Description="Deletion of mobile element">\n'
    '##ALT=<ID=INS:ME,Description="Insertion of mobile element">\n'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')


class VcfWriter:
    def __init__(self, fname):
        try:
            self._out = open(fname, "w+b")         # (readable too: map_region maps spans of it)
        except IOError as e:
            raise VcfWriterError(f"Cannot write to VCF file {fname} {e}")

    def __del__(self):
        self.close()

    def close(self):
        out = getattr(self, "_out", None)
        if out is not None and not out.closed:
            out.flush()
            if os.fstat(out.fileno()).st_size > out.tell():     # (a mapped region that was cut short)
                os.ftruncate(out.fileno(), out.tell())
            out.close()

    def write_header(self, input_fasta, fasta, assembly_name: str, species_name: str,
                     sample_name: str):
        now = datetime.now()
        lines = ["##fileformat=VCFv4.3\n",
                 f"##filedate={now.year}{now.month}{now.day}\n",      # unpadded, as the reference
                 "##source=Mutation-Simulator\n",
                 f"##reference={input_fasta}\n"]
        if hasattr(fasta, "names_and_lengths"):          # (our loader: no record object per contig of an assembly)
            pairs = fasta.names_and_lengths()
        else:
            pairs = ((fasta[key].name, len(fasta[key])) for key in list(fasta.keys()))
        for name, length in pairs:
            lines.append(f"##contig=<ID={name},length={length},assembly={assembly_name},"
                         f"species=\"{species_name}\">\n")
        lines.append(_HEADER_TAIL)
        lines.append(f"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{sample_name}\n")
        self._out.write("".join(lines).encode("utf-8", "replace"))

    def write(self, record: VcfRecord, seq_name: str):
        if record.ref != record.alt:
            self._out.write(f"{seq_name}\t{record.start}\t.\t{record.ref}\t{record.alt}\t.\t.\t"
                            f"{record.info}\tGT\t1\n".encode("utf-8", "replace"))

    def write_raw(self, text: bytes):
        self._out.write(text)

    def map_region(self, nbytes: int):
        """The next ``nbytes`` of the file mapped for writing (record lines rendered elsewhere land there directly)."""
        from .fasta_writer import MappedRegion
        return MappedRegion(self._out, nbytes)

    def commit_region(self, region):
        region.close(self._out)

    def native_span(self):
        """(fd, position) where the next record lines go, for a writer outside Python (``msim_render_vcf_device_file``);
        ``commit_native(pos, nbytes)`` finishes it."""
        self._out.flush()
        return self._out.fileno(), self._out.tell()

    def commit_native(self, pos: int, nbytes: int):
        self._out.seek(pos + nbytes)

    def tell(self) -> int:
        self._out.flush()
        return self._out.tell()
