"""Simulation settings: ARGS flags or an RMT file -> per-contig range descriptors.

Host-side mirror of the reference's settings model (rmt.py:79-776): same class and exception
names, same validation order, same messages, same quirks (they decide which numbers reach the
kernels, so they are part of parity):

  * a TL rate is split in two halves, TL and TLI, *after* validation (rmt.py:91-94);
  * chances keep dict insertion order -- ARGS: SN, IN, DE, IV, DU, TL, TLI; RMT: token order on
    the line, TLI last (rmt.py:443-450, 676-692) -- because ``numpy.random.choice`` walks that order;
  * every non-SN type *present* in the rate dict is length-checked, even at rate 0 (rmt.py:119-137);
  * the whole RMT text is lower-cased, meta values included (rmt.py:584);
  * ``END`` is resolved only on the last listed range of a contig (rmt.py:368-374);
  * gaps are filled with the standard settings, overlaps yield negative-length fillers
    (rmt.py:222-258) that later make ``random.sample`` raise ``ValueError``.

Everything floating-point on the path (rate sums, chances, titv) is evaluated here, in Python,
with the reference's own expressions; ``plan_descriptors()`` turns it into the integers the C-ABI
takes (include/msim.h).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

from .defaults import Defaults
from .mut_types import MutType
from .util import print_warning

_IND = ("sn", "in", "de", "iv", "du", "tl")
MUT_INDICATORS = set(_IND)
MAX_LENG_INDICATORS = {f"{i}max" for i in _IND if i != "sn"}
MIN_LENG_INDICATORS = {f"{i}min" for i in _IND if i != "sn"}
META_KEYWORDS_STR = {"fasta", "md5", "species_name", "assembly_name", "sample_name"}
META_KEYWORDS_FLOAT = {"titv"}
META_KEYWORDS_BLOCK = {f"{i}_block" for i in _IND}


class RatesTooHighError(Exception):
    pass


class RatesTooLowError(Exception):
    pass


class ItRateTooHighError(Exception):
    pass


class ItRateTooLowError(Exception):
    pass


class ITNotEnoughAvailChromsError(Exception):
    pass


class TitvTooLowError(Exception):
    pass


class RMTParseError(Exception):
    pass


class MissingLengthError(Exception):
    pass


class MinimumLengthTooLowError(Exception):
    pass


class MinimumLengthHigherThanMaximumError(Exception):
    pass


class ChromNotExistError(Exception):
    pass


class RangeDefinitionOutOfBoundsError(Exception):
    pass


class MutationSettings:
    """Rates, derived chances and length bounds of one range (reference rmt.py:79-163)."""

    def __init__(self, mut_rates: Optional[dict], mut_lengs: Optional[dict]):
        self.mut_rates = mut_rates
        self.mut_lengs = mut_lengs
        self._check()
        if self.mut_rates and MutType.TL in self.mut_rates:
            half = self.mut_rates[MutType.TL] / 2
            self.mut_rates[MutType.TL] = half
            self.mut_rates[MutType.TLI] = half
        if self.mut_rates:
            total = sum(self.mut_rates.values())
            self.mut_chances = {t: r / total for t, r in self.mut_rates.items()}
        else:
            self.mut_chances = None

    def _check(self):
        rates = self.mut_rates
        if not rates:
            return
        if any(r < 0 for r in rates.values()) or sum(rates.values()) <= 0:
            low = ", ".join(t.name for t, r in rates.items() if r <= 0)
            raise RatesTooLowError(f"Mutation rate/s too low for: {low}. If this was intentional "
                                   "use the None keyword instead")
        if sum(rates.values()) > 0.5:
            above_one = ", ".join(t.name for t, r in rates.items() if r > 1)
            if above_one:
                raise RatesTooHighError(f"Mutation rate/s too high for: {above_one}")
            raise RatesTooHighError("Sum of mutation rates too high")
        lengs = self.mut_lengs
        for t in rates:
            if t is MutType.SN:
                continue
            if not lengs or t not in lengs["min"] or t not in lengs["max"]:
                raise MissingLengthError(f"Missing length keyword for: {t.name}")
            lo, hi = lengs["min"][t], lengs["max"][t]
            if lo > hi:
                raise MinimumLengthHigherThanMaximumError(
                    f"Minimum length is greater and maximum length for: {t.name}")
            if lo < (2 if t is MutType.IV else 1):
                raise MinimumLengthTooLowError(f"Minimum length too low for: {t.name}")

    def __repr__(self) -> str:
        return f"Rates: {self.mut_rates}, Chances: {self.mut_chances}, Lengs: {self.mut_lengs}"

    @property
    def has_mutations(self) -> bool:
        return bool(self.mut_rates and any(r for r in self.mut_rates.values()))


class RangeDefinition:
    """0-based inclusive [start, stop] plus the settings that apply there (rmt.py:166-189)."""

    def __init__(self, start: int, stop, mutation_settings: MutationSettings):
        self.start = start
        self.stop = stop
        self.mutation_settings = mutation_settings

    def __repr__(self) -> str:
        return f"{self.start}-{self.stop} {self.mutation_settings}"

    def eval_end(self, chrom_length: int):
        if self.stop == "end":
            self.stop = chrom_length - 1


class ChromosomeSettings:
    """Range definitions (and the it rate) of one contig (rmt.py:192-258)."""

    def __init__(self, number: int, it_rate: Optional[float], range_definitions: list):
        self.number = number
        self.it_rate = it_rate
        if it_rate and it_rate > 0.5:
            raise ItRateTooHighError("Interchromosomal translocation rate too high")
        if it_rate and it_rate < 0:
            raise ItRateTooLowError("Interchromosomal translocation rate too low")
        self.range_definitions = range_definitions

    def __repr__(self) -> str:
        return f"{self.number}\nit={self.it_rate}\n{self.range_definitions}\n"

    def fill_missing_ranges(self, chrom_length: int, std: MutationSettings):
        rds = self.range_definitions
        if not rds:
            rds.append(RangeDefinition(0, chrom_length - 1, std))
            return
        if rds[0].start != 0:
            rds.insert(0, RangeDefinition(0, rds[0].start - 1, std))
        if rds[-1].stop != chrom_length - 1:
            rds.append(RangeDefinition(rds[-1].stop + 1, chrom_length - 1, std))
        i = 0
        while i < len(rds) - 1:          # one pass; an inserted filler is itself contiguous
            if rds[i].stop + 1 != rds[i + 1].start:
                rds.insert(i + 1, RangeDefinition(rds[i].stop + 1, rds[i + 1].start - 1, std))
            i += 1


class StdChromosomes:
    """``SimulationSettings.chromosomes`` when no contig has range definitions of its own (ARGS mode): contig i is ONE
    range [0, len_i - 1] with the std settings (what ``_fill_missing_chroms`` would append, rmt.py:353-359).  The
    ``ChromosomeSettings`` objects are made on demand -- an assembly has 10^5 contigs and the batch path of the mutation
    pass never looks at them one by one -- and behave like the list they replace (len, index, slice, iteration)."""

    def __init__(self, lengths, std: MutationSettings, std_it):
        self.lengths = lengths
        self.std = std
        self.std_it = std_it

    def __len__(self) -> int:
        return len(self.lengths)

    def _make(self, i: int) -> ChromosomeSettings:
        return ChromosomeSettings(i, self.std_it, [RangeDefinition(0, int(self.lengths[i]) - 1, self.std)])

    def __getitem__(self, key):
        if isinstance(key, slice):
            return [self._make(i) for i in range(*key.indices(len(self)))]
        n = len(self)
        if key < 0:
            key += n
        if not 0 <= key < n:
            raise IndexError("chromosome index out of range")
        return self._make(key)

    def __iter__(self):
        return (self._make(i) for i in range(len(self)))

    def __repr__(self) -> str:
        return repr(list(self))


class SimulationSettings:
    """Everything the mutation pass needs (rmt.py:261-776)."""

    def __init__(self, std: MutationSettings, std_it: Optional[float], chromosomes: list,
                 mut_block: Optional[dict], fasta=None, md5: Optional[str] = None,
                 titv: float = Defaults.TITV, species_name: str = Defaults.SPECIES_NAME,
                 assembly_name: str = Defaults.ASSEMBLY_NAME,
                 sample_name: str = Defaults.SAMPLE_NAME,
                 ignore_warnings: bool = Defaults.IGNORE_WARNINGS,
                 no_color: bool = Defaults.NO_COLOR):
        self._std = std
        self._std_it = std_it
        self.chromosomes = chromosomes
        self.mut_block = mut_block
        self._normalise_blocks(ignore_warnings, no_color)
        self.fasta = fasta
        self.md5 = md5
        self.titv = titv
        if self.titv < 0:
            raise TitvTooLowError("Titv value is below 0")
        self.species_name = species_name
        self.assembly_name = assembly_name
        self.sample_name = sample_name

    def __repr__(self) -> str:
        return (f"[META]\nfasta={self.fasta}\nmd5={self.md5}\ntitv={self.titv}\n"
                f"species_name={self.species_name}\nassembly_name={self.assembly_name}\n"
                f"sample_name={self.sample_name}\nmut_block={self.mut_block}\n\n[STD]\n"
                f"it={self._std_it}\n{self._std}\n\n[RD]\n{self.chromosomes}")

    # ------------------------------------------------------------------ validation helpers
    def _normalise_blocks(self, ignore_warnings: bool, no_color: bool):
        if not self.mut_block:
            self.mut_block = Defaults.MUT_BLOCK
            return
        for t in MutType:
            if t is MutType.TLI:
                continue
            if t not in self.mut_block:
                self.mut_block[t] = 1
            elif self.mut_block[t] < 1:
                self.mut_block[t] = 1
                if not ignore_warnings:
                    print_warning(f"'{t.name}' block was set to 1", no_color)
        self.mut_block[MutType.TLI] = self.mut_block[MutType.TL]

    def _validate_it(self, fasta):
        usable = [c.number for c in self.chromosomes
                  if c.it_rate is not None and len(fasta[c.number]) > 2]
        if len(usable) < 2:
            raise ITNotEnoughAvailChromsError(
                "Not enought available chromosomes for interchromosomal translocations")
        if sum(self.chromosomes[n].it_rate for n in usable) == 0:
            raise ItRateTooLowError("Interchromosomal translocation rates are too low")

    def _check_chroms_exist(self, fasta):
        names = list(fasta.keys())
        for c in self.chromosomes:
            try:
                names[c.number]
            except IndexError:
                raise ChromNotExistError(
                    f"Chromosome {c.number+1} does not exist in the fasta file")

    def _eval_chrom_ends(self, fasta):
        for c in self.chromosomes:
            if c.range_definitions:
                c.range_definitions[-1].eval_end(len(fasta[c.number]))

    def _fill_missing_chroms(self, fasta):
        listed = {c.number for c in self.chromosomes}
        for idx in range(len(list(fasta.keys()))):
            if idx not in listed:
                whole = RangeDefinition(0, len(fasta[idx]) - 1, self._std)
                self.chromosomes.append(ChromosomeSettings(idx, self._std_it, [whole]))

    def _sort(self):
        self.chromosomes = sorted(self.chromosomes, key=lambda c: c.number)
        for c in self.chromosomes:
            c.range_definitions = sorted(c.range_definitions, key=lambda rd: rd.start)

    def _check_bounds(self, fasta):
        names = list(fasta.keys())
        for c in self.chromosomes:
            if not c.range_definitions:
                continue
            if c.range_definitions[0].start < 0:
                raise RangeDefinitionOutOfBoundsError(
                    f"A range definition of chromosome {c.number+1} is starting at 0")
            if c.range_definitions[-1].stop > len(fasta[names[c.number]]):
                raise RangeDefinitionOutOfBoundsError(
                    f"A range definition of chromosome {c.number+1} is longer than the chromosome")

    def _fill_missing_chrom_ranges(self, fasta):
        for c in self.chromosomes:
            c.fill_missing_ranges(len(fasta[c.number]), self._std)

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_args(cls, args, fasta, ignore_warnings: bool) -> "SimulationSettings":
        T = MutType
        rates = {T.SN: args.snp, T.IN: args.insert, T.DE: args.deletion, T.IV: args.inversion,
                 T.DU: args.duplication, T.TL: args.translocation}
        lengs = {
            "min": {T.IN: args.insertminlength, T.DE: args.deletionminlength,
                    T.IV: args.inversionminlength, T.DU: args.duplicationminlength,
                    T.TL: args.translocationminlength},
            "max": {T.IN: args.insertmaxlength, T.DE: args.deletionmaxlength,
                    T.IV: args.inversionmaxlength, T.DU: args.duplicationmaxlength,
                    T.TL: args.translocationmaxlength}}
        block = {T.SN: args.snpblock, T.IN: args.insertblock, T.DE: args.deletionblock,
                 T.IV: args.inversionblock, T.DU: args.duplicationblock,
                 T.TL: args.translocationblock}
        sim = cls(MutationSettings(rates, lengs), None, [], block,
                  titv=args.transitionstransversions, species_name=args.species,
                  assembly_name=args.assembly, sample_name=args.sample,
                  ignore_warnings=ignore_warnings)
        # no contig is listed: every one is the std range over its whole length (lazily, see StdChromosomes)
        table = getattr(fasta, "index_table", None)
        lengths = (table["n_bases"] if table is not None and len(table) == len(fasta.keys())
                   else [len(fasta[i]) for i in range(len(list(fasta.keys())))])
        sim.chromosomes = StdChromosomes(lengths, sim._std, sim._std_it)
        return sim

    @classmethod
    def from_it(cls, it_rate: float, fasta, ignore_warnings: bool) -> "SimulationSettings":
        sim = cls(MutationSettings(None, None), it_rate, [], None, ignore_warnings=ignore_warnings)
        sim._fill_missing_chroms(fasta)
        sim._sort()
        sim._validate_it(fasta)
        return sim

    @classmethod
    def from_rmt(cls, path, fasta, ignore_warnings: bool) -> "SimulationSettings":
        sections = _split_sections(_read_rmt_lines(path))
        if len(sections["std"]) != 2:
            raise RMTParseError(
                f"Standard section not defined or malformed. Occurred while reading {path}")
        try:
            meta, block = _parse_meta(sections["meta"])
            std_it = _parse_it(sections["std"][0])
            std = _parse_settings(sections["std"][1])
            chroms = _parse_rd(sections["rd"], std_it)
            sim = cls(std, std_it, chroms, block, ignore_warnings=ignore_warnings, **meta)
            sim._check_chroms_exist(fasta)
            sim._eval_chrom_ends(fasta)
            sim._fill_missing_chroms(fasta)
            sim._sort()
            sim._check_bounds(fasta)
            sim._fill_missing_chrom_ranges(fasta)
            if sim.has_it:
                sim._validate_it(fasta)
        except Exception as e:
            raise type(e)(f"{e}. Occurred while reading {path}")
        return sim

    # ------------------------------------------------------------------ queries
    @property
    def has_mutations(self) -> bool:
        if isinstance(self.chromosomes, StdChromosomes):
            return len(self.chromosomes) > 0 and self._std.has_mutations
        return any(rd.mutation_settings.has_mutations
                   for c in self.chromosomes for rd in c.range_definitions)

    @property
    def has_it(self) -> bool:
        if isinstance(self.chromosomes, StdChromosomes):
            return len(self.chromosomes) > 0 and bool(self._std_it)
        return any(c.it_rate for c in self.chromosomes)


# ---------------------------------------------------------------------- RMT text -> pieces
def _read_rmt_lines(path) -> list:
    kept = []
    with open(path, "r") as fh:
        for line in fh.readlines():
            if line.startswith("#"):
                continue
            line = line.split("#")[0].strip()
            if line:
                kept.append(line)
    return kept


def _split_sections(lines: list) -> dict:
    out = {"meta": [], "std": [], "rd": []}
    where = "meta"
    for line in lines:
        line = line.lower()
        if line == "std":
            where = "std"
            continue
        if line.startswith("chr"):
            where = "rd"
        out[where].append(line)
    return out


def _parse_meta(lines: list):
    meta, block = {}, {}
    for line in lines:
        key, val = [tok.strip() for tok in line.split("=")]
        if key in META_KEYWORDS_STR:
            meta[key] = val
        elif key in META_KEYWORDS_FLOAT:
            try:
                meta[key] = float(val)
            except ValueError:
                raise RMTParseError(
                    f"{key.capitalize()} value of '{val}' is not representable as a float")
        elif key in META_KEYWORDS_BLOCK:
            try:
                block[MutType[key.removesuffix("_block").upper()]] = int(val)
            except ValueError:
                raise RMTParseError(f"Mut block value of {key} is not representable as an integer")
    return meta, block


def _parse_it(line: str) -> Optional[float]:
    toks = [t.strip() for t in line.split(" ") if t.strip()]
    if toks[0] != "it" or len(toks) != 2:
        raise RMTParseError("Malformed interchromosomal translocation rate setting")
    if toks[1] == "none":
        return None
    try:
        return float(toks[1])
    except ValueError:
        raise RMTParseError("Malformed interchromosomal translocation rate setting")


def _parse_settings(line: str) -> MutationSettings:
    if line == "none":
        return MutationSettings(None, None)
    toks = [t.strip() for t in line.split(" ") if t.strip()]
    if len(toks) % 2:
        raise RMTParseError("Malformed mutation settings")
    rates, lengs = {}, {"min": {}, "max": {}}
    for i in range(len(toks) - 1):       # every token but the last may be a keyword
        tok = toks[i]
        try:
            if tok in MUT_INDICATORS:
                rates[MutType[tok.upper()]] = float(toks[i + 1])
            elif tok in MAX_LENG_INDICATORS:
                lengs["max"][MutType[tok.removesuffix("max").upper()]] = int(toks[i + 1])
            elif tok in MIN_LENG_INDICATORS:
                lengs["min"][MutType[tok.removesuffix("min").upper()]] = int(toks[i + 1])
        except ValueError:
            raise RMTParseError("Malformed mutation settings")
    return MutationSettings(rates, lengs)


def _parse_range(text: str):
    toks = [t.strip() for t in text.split("-") if t.strip()]
    if len(toks) != 2 or text.count("-") != 1:
        raise RMTParseError("Malformed range in range definitions")
    try:
        start = int(toks[0]) - 1
        stop = toks[1] if toks[1] == "end" else int(toks[1]) - 1
    except ValueError:
        raise RMTParseError("Malformed range in range definitions")
    return start, stop


def _parse_rd(rows: list, std_it: Optional[float]) -> list:
    table: dict = {}
    current = None
    for row in rows:
        if row.startswith("chr"):
            try:
                current = int(row.split(" ")[-1]) - 1
            except ValueError:
                raise RMTParseError(f"Chromosome index {row} is invalid")
            table[current] = {"range_definitions": []}
        elif row.startswith("it"):
            table[current]["it"] = _parse_it(row)
        else:
            span = row.partition(" ")[0]
            start, stop = _parse_range(span)
            settings = _parse_settings(row.removeprefix(span + " "))
            table[current]["range_definitions"].append(RangeDefinition(start, stop, settings))
    return [ChromosomeSettings(idx, entry.get("it", std_it), entry["range_definitions"])
            for idx, entry in table.items()]
