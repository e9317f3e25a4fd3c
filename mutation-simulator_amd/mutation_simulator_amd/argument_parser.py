"""Command line of ``mutation-simulator`` -- flag-for-flag the reference's ``args`` and ``rmt``
sub-commands (reference argument_parser.py:31-240) plus a few additions of ours that never change
a default: ``--seed``, ``--device``, ``--gpus``, ``--rng``, ``--bench-json``.

The ``it`` sub-command (inter-chromosomal translocations, reference it_mutator.py: a second pass over
the Fasta) runs through ``ITMutator`` / ``BedpeWriter``.  ``--rng fast`` applies to the mutation pass
(``args`` / ``rmt``) only: the IT pass always draws from CPython's generator.
"""
from __future__ import annotations

from argparse import ArgumentParser, Namespace
from pathlib import Path

from ._version import __version__
from .defaults import Defaults as D


def add_outfile_names(args: Namespace) -> Namespace:
    """``<outbase>_ms<infile suffix>`` / ``<outbase>_ms.vcf`` (reference argument_parser.py:14-28).

    ``-o`` may be a basename or a directory-like path with an empty stem (".", "dir/.."): then the
    input's stem is used inside that directory.
    """
    try:
        args.outbase = args.outbase.with_stem(args.outbase.stem + "_ms")
    except ValueError:
        args.outbase = args.outbase / (args.infile.stem + "_ms")
    args.outfasta = args.outbase.with_suffix(args.infile.suffix)
    args.outfastait = args.outfasta.with_stem(args.outfasta.stem + "_it")
    args.outvcf = args.outfasta.with_suffix(".vcf")
    args.outbedpe = args.outfastait.with_suffix(".bedpe")
    return args


# (short, long, kind, default, help) for one mutation type: rate / min / max / block
def _type_options(short: str, long: str, label: str, plural: str, minlen, maxlen, after: str):
    opts = [(f"-{short}", f"--{long}", float, D.RATE, f"{label} rate. Default = {D.RATE}")]
    if minlen is not None:
        opts.append((f"-{short}min", f"--{long}minlength", int, minlen,
                     f"Minimum length of {plural}. Default = {minlen}"))
        opts.append((f"-{short}max", f"--{long}maxlength", int, maxlen,
                     f"Maximum length of {plural}. Default = {maxlen}"))
    opts.append((f"-{short}b", f"--{long}block", int, D.BLOCK,
                 f"Amount of bases blocked after {after}. Default = {D.BLOCK}"))
    return opts


def build_parser() -> ArgumentParser:
    parser = ArgumentParser(
        prog="mutation-simulator",
        description="See https://github.com/mkpython3/Mutation-Simulator for more information "
                    "about this program.")
    parser.add_argument("infile", type=Path, help="Path of the reference Fasta file")
    parser.add_argument("-o", "--output", type=Path, default=D.OUTBASE, dest="outbase",
                        help="Path/Basename for the output files (without file extension)")
    for flag, long, default, text in (
            ("-w", "--ignore-warnings", D.IGNORE_WARNINGS, "Silences warnings"),
            ("-c", "--no-color", D.NO_COLOR, "Always disable color"),
            ("-p", "--no-progress", D.NO_PROGRESS, "Disable progressbars"),
            ("-q", "--quiet", D.QUIET, "Disable all output except errors")):
        parser.add_argument(flag, long, action="store_true", default=default, help=text)
    parser.add_argument("-v", "--version", action="version",
                        version=f"Mutation-Simulator {__version__}")
    # additions of this build (defaults keep the reference's behaviour)
    parser.add_argument("--seed", type=int, default=None,
                        help="Seed both random streams (random.seed(S); numpy.random.seed(S)) "
                             "for reproducible output")
    parser.add_argument("--device", type=int, default=0, help="GPU ordinal to run on")
    parser.add_argument("--gpus", type=int, default=1,
                        help="Shard the contigs over this many GPUs of the node (one worker process per GPU; output is "
                             "byte-identical to a 1-GPU run). Default = 1")
    parser.add_argument("--rng", choices=["compat", "fast"], default="compat",
                        help="compat (default): the reference's two MT19937 streams, output bit-identical to the reference "
                             "under the same seeds. fast: a counter-based generator (Philox) -- same distributions, NOT the "
                             "reference's numbers; no sequential chain in the draw; SNP-only settings")
    parser.add_argument("--bench-json", type=Path, default=None,
                        help="Write per-stage timings of the mutation pass to this JSON file")

    sub = parser.add_subparsers(
        dest="mode",
        help="Generate mutations or interchromosomal translocations via RMT or arguments")
    sub.required = True

    p_args = sub.add_parser("args", help="Use commandline arguments for mutations instead of RMT")
    table = _type_options("sn", "snp", "SNP", "", None, None, "SNP")
    table.insert(2, ("-titv", "--transitionstransversions", float, D.TITV,
                     f"Ratio of transitions:transversions likelihood. Default = {D.TITV}"))
    table += _type_options("in", "insert", "Insert", "inserts", D.MINLEN, D.MAXLEN, "insert")
    table += _type_options("de", "deletion", "Deletion", "deletions", D.MINLEN, D.MAXLEN,
                           "deletion")
    table += _type_options("iv", "inversion", "Inversion", "inversion", D.IV_MINLEN, D.IV_MAXLEN,
                           "inversion")
    table += _type_options("du", "duplication", "Duplication", "duplications", D.MINLEN,
                           D.MAXLEN, "duplication")
    table += _type_options("tl", "translocation", "Translocation", "translocations", D.MINLEN,
                           D.MAXLEN, "translocations")
    for short, long, kind, default, text in table:
        p_args.add_argument(short, long, type=kind, default=default, help=text)
    for short, long, default, what in (("-a", "--assembly", D.ASSEMBLY_NAME, "Assembly"),
                                       ("-s", "--species", D.SPECIES_NAME, "Species"),
                                       ("-n", "--sample", D.SAMPLE_NAME, "Sample")):
        p_args.add_argument(short, long, default=default,
                            help=f"{what} name for the VCF file. Default = '{default}'")

    p_it = sub.add_parser("it", help="Generate interchromosomal translocations via the command line")
    p_it.add_argument("interchromosomalrate", type=float,
                      help="Rate of interchromosomal translocations")

    p_rmt = sub.add_parser("rmt", help="Use random mutation table instead of arguments")
    p_rmt.add_argument("rmtfile", type=Path, help="Path to the RMT file")
    return parser


def get_args(argv=None) -> Namespace:
    args = build_parser().parse_args(argv)
    if args.quiet:
        args.ignore_warnings = True
        args.no_progress = True
    return add_outfile_names(args)
